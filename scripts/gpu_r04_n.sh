#!/bin/bash
# round 4: transpose-popcount flush of the classification counters -- class / list / variant tests, single-call probe
TAG=${1:-r04n}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py -q -m gpu > $OUT/t_classes.log 2>&1; tail -4 $OUT/t_classes.log
timeout 600 python scripts/probe_single_pass.py 10000 5000000 once > $OUT/probe.log 2>&1; cat $OUT/probe.log
