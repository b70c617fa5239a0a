#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
cd /tmp
TRACS_BENCH_PARTIAL=0.005 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_partial -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace_partial.log 2>&1
rm -f $OUT/trace_partial/trace_kernel_trace.csv
grep -E "pairsnp_mfma|general_fixup" $OUT/trace_partial/trace_kernel_stats.csv | cut -c1-60,300-420
cd $GRAFT_REPO_ROOT
for st in 4x8 8x8 8x16 2x4; do
  echo "== supertile $st" | tee -a $OUT/supertile.log
  TRACS_SUPERTILE=$st timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tee -a $OUT/supertile.log
done
