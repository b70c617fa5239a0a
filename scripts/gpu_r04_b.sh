#!/bin/bash
# round 4: the rebuilt lists (site_lists.hip) -- structure tests, site-class parity, then the probe and the bench line
TAG=${1:-r04b}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py -x -q -m gpu > $OUT/t_lists.log 2>&1; tail -25 $OUT/t_lists.log
timeout 1200 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/t_classes.log 2>&1; tail -25 $OUT/t_classes.log
timeout 600 python scripts/probe_single_pass.py > $OUT/probe.log 2>&1; cat $OUT/probe.log
