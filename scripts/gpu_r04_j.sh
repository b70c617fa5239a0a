#!/bin/bash
# round 4: merged line ring / cheaper scan, listed counts from the per-site pass -- list tests, segment probe, single-pass probe
TAG=${1:-r04j}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py -q -m gpu > $OUT/t_lists.log 2>&1; tail -4 $OUT/t_lists.log
timeout 600 python scripts/probe_nn_segments.py 10000 5000000 0,512,256,128 > $OUT/segments.log 2>&1; cat $OUT/segments.log
timeout 600 python scripts/probe_single_pass.py 10000 5000000 once > $OUT/probe.log 2>&1; cat $OUT/probe.log
