#!/bin/bash
# round 4: bucketed fill of the per-sample listed entries -- list / class / config tests, the bench workload's probe, the partial-code workload's trace
TAG=${1:-r04ab}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_configs.py -q -m gpu > $OUT/t.log 2>&1; tail -6 $OUT/t.log
timeout 600 python scripts/probe_single_pass.py 10000 5000000 once > $OUT/probe.log 2>&1; tail -4 $OUT/probe.log
bash scripts/gpu_r04_general.sh $TAG
