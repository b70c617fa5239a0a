#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for st in 4x8 8x8 2x4; do
  echo "== general 0.5% partial, supertile $st" | tee -a $OUT/general.log
  TRACS_SUPERTILE=$st TRACS_BENCH_PARTIAL=0.005 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/general.log
done
echo "== consensus" | tee -a $OUT/general.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/general.log
