#!/bin/bash
# round 2, first GPU pass: parity tests on the new matrix-core kernels, then a tile-shape sweep at 10 000 x 400 kbp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
for shape in 3x2 2x2 2x3 4x2; do
  echo "== consensus $shape" >> gpurun_out/r02a/sweep.log
  TRACS_MFMA_TILE=$shape TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02a/sweep.log 2>&1
  echo "== general(forced) $shape" >> gpurun_out/r02a/sweep.log
  TRACS_FORCE_GENERAL=1 TRACS_MFMA_TILE=$shape TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02a/sweep.log 2>&1
done
echo "== general with 0.5% partial codes, default shape" >> gpurun_out/r02a/sweep.log
TRACS_BENCH_PARTIAL=0.005 TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02a/sweep.log 2>&1
echo "== general with 0.5% partial codes, VALU kernel" >> gpurun_out/r02a/sweep.log
TRACS_GENERAL_MFMA=0 TRACS_BENCH_PARTIAL=0.005 TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02a/sweep.log 2>&1
grep -E "^==|kernel_ms" gpurun_out/r02a/sweep.log | sed -e 's/.*"kernel_ms": \([0-9.]*\).*/   kernel_ms \1/' 
