#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
for pf in 1 0 1 0; do
  echo "== consensus prefetch=$pf" | tee -a $OUT/prefetch.log
  TRACS_MFMA_PREFETCH=$pf timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/prefetch.log
done
for pf in 1 0; do
  echo "== general 0.5% prefetch=$pf" | tee -a $OUT/prefetch.log
  TRACS_MFMA_PREFETCH=$pf TRACS_BENCH_PARTIAL=0.005 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/prefetch.log
done
