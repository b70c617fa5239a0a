#!/bin/bash
# round 2: tile-shape sweep at 10 000 x 400 kbp after spreading the staging loads through the units
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
for shape in 3x2 2x2 2x3; do
  echo "== consensus $shape" >> gpurun_out/r02b/sweep.log
  TRACS_MFMA_TILE=$shape TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02b/sweep.log 2>&1
  echo "== general 0.5% partial $shape" >> gpurun_out/r02b/sweep.log
  TRACS_BENCH_PARTIAL=0.005 TRACS_MFMA_TILE=$shape TRACS_BENCH_SITES=400000 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/r02b/sweep.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02b/sweep.log | sed -e 's/.*"kernel_ms": \([0-9.]*\).*/   kernel_ms \1/'
# kernel trace of the general path (which kernels, how long)
cd /tmp && TRACS_BENCH_PARTIAL=0.005 TRACS_MFMA_TILE=2x2 TRACS_BENCH_SITES=400000 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02b/trace_general -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02b/trace_general.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r02b/trace_general -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
find gpurun_out/r02b/trace_general -name "*.db" -delete 2>/dev/null; find gpurun_out/r02b/trace_general -name "*kernel_trace.csv" -size +5M -delete
