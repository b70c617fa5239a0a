#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ai
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
rm -f $OUT/trace/trace_kernel_trace.csv
grep -E "plane_popcount|compact_sites" $OUT/trace/trace_kernel_stats.csv | cut -c1-60,180-300
