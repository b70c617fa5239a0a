#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
TRACS_MFMA_TILE=2x2w4x2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for shape in 2x2 2x2w4x2 2x2w2x4; do
  echo "== consensus $shape" | tee -a $OUT/waves.log
  TRACS_MFMA_TILE=$shape timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/waves.log
  echo "== general 0.5% $shape" | tee -a $OUT/waves.log
  TRACS_MFMA_TILE=$shape TRACS_BENCH_PARTIAL=0.005 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/waves.log
done
