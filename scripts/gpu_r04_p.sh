#!/bin/bash
# round 4: site-sharded bench with the result left distributed -- the scale tests (two gloo ranks on the GPU, RCCL world 1, CLI)
TAG=${1:-r04p}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -x > $OUT/t_scale.log 2>&1; tail -15 $OUT/t_scale.log
