#!/bin/bash
# usage: bash scripts/gpu_trace.sh <tag>  -- bench line, rocprofv3 kernel trace of the SAME command, PMC traffic
TAG=${1:-t01}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python bench.py --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_c3.json
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 10 --warmup 2 2>&1 | tail -1 | tee $OUT/bench_c2.json
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
rm -f $OUT/trace/trace_kernel_trace.csv
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py 10000 5000000 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-include-regex "pairsnp_" --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $T > $OUT/pmc_$c.log 2>&1
  grep -h "pairsnp" $OUT/pmc_$c/pmc_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | tr -d '"' | tee -a $OUT/pmc_traffic.txt
done
