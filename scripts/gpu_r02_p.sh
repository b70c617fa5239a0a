#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp
TRACS_CONS_WORDS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
for cfg in "0 2x2" "1 2x2" "1 2x2w4x2" "0 2x2" "1 2x2"; do
  set -- $cfg
  echo "== consensus words=$1 shape=$2" | tee -a $OUT/words.log
  TRACS_CONS_WORDS=$1 TRACS_MFMA_TILE=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/words.log
done
