#!/bin/bash
# round 3 artefacts: driver-style bench line, rocprofv3 kernel trace of the same command, FETCH_SIZE / WRITE_SIZE of the
# list walk and the counting pass from bench.py itself (separate --pmc passes, program directly after --), config 4 at its shape.
# usage (GPU box): bash scripts/gpu_r03_artifacts.sh <tag>
TAG=${1:-r03a}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json; cut -c1-300 $OUT/bench_c3.json
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1
cp /tmp/$TAG/trace/*/trace_kernel_stats.csv $OUT/c3_kernel_stats.csv 2>/dev/null || cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/c3_kernel_stats.csv
head -8 $OUT/c3_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-include-regex "nn_rows|pairsnp_mfma|general_fixup" --pmc $c --output-format csv -d /tmp/$TAG/pmc_$c -o pmc -- $B --steps 1 --warmup 0 > $OUT/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-include-regex "nn_rows|pairsnp_mfma" --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/$TAG/pmc_l2 -o pmc -- $B --steps 1 --warmup 0 > $OUT/pmc_l2.log 2>&1
python3 - <<PY > $OUT/pmc_bench.txt
import csv, collections, glob
for d in ['pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_l2']:
    fs = glob.glob('/tmp/$TAG/' + d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(d, 'missing'); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, '|', k, '|', {c: ("%.6g" % (sum(x) / len(x)), len(x)) for c, x in v.items()})
PY
cat $OUT/pmc_bench.txt
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/bench_config4.py > $OUT/bench_config4.json 2> $OUT/bench_config4.log; cut -c1-400 $OUT/bench_config4.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/c4 -o trace -- python3 $GRAFT_REPO_ROOT/scripts/bench_config4.py --samples 10000 > $OUT/c4_trace.log 2>&1
cp $(find /tmp/$TAG/c4 -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv; grep -E "posterior" $OUT/c4_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/bench_config5.py > $OUT/bench_config5.json 2> $OUT/bench_config5.log; cut -c1-400 $OUT/bench_config5.json
