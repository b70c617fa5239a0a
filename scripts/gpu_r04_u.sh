#!/bin/bash
TAG=${1:-r04u}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lists.py -q -m gpu > $OUT/t_lists.log 2>&1; tail -15 $OUT/t_lists.log
