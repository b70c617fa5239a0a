#!/bin/bash
# round 6 artefacts: driver-style bench line (with the filter leg, the per-call sensitivity legs and the coverage workload), rocprofv3 kernel
# trace of the same command, FETCH_SIZE / WRITE_SIZE / L2 of the list kernels, the transcluster gather and the filter's pair kernel from
# bench.py itself (separate --pmc passes, program directly after --), config 2, the partial-code alignment, the coverage workload's
# kernel trace, the command line end to end, the N = 2 line through gloo.   usage (GPU box): bash scripts/gpu_r06_artifacts.sh <tag>
TAG=${1:-r06art}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json; cut -c1-300 $OUT/bench_c3.json
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1
cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_c3_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/bench_c3_kernel_stats.csv")):
    if "tracs::" in r["Name"]:
        print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
RX="nn_rows_kernel|site_lists_kernel|classify_sites|n_bitmap_kernel|minor_fixup_kernel|tc_ratio_keys|tc_table_gather2|tc_mark"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-include-regex "$RX" --pmc $c --output-format csv -d /tmp/$TAG/pmc_$c -o pmc -- $B --steps 1 --warmup 0 > $OUT/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-include-regex "$RX" --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/$TAG/pmc_l2 -o pmc -- $B --steps 1 --warmup 0 > $OUT/pmc_l2.log 2>&1
python3 - <<PY > $OUT/pmc_bench_c3.txt
import csv, collections, glob
for d in ['pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_l2']:
    fs = glob.glob('/tmp/$TAG/' + d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(d, 'missing'); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, '|', k, '|', {c: ("%.6g" % (sum(x) / len(x)), len(x)) for c, x in v.items()})
PY
cat $OUT/pmc_bench_c3.txt
# the coverage workload (what tracs align writes): kernel trace of two calls
WORKLOAD=coverage timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/cov -o trace -- python3 $GRAFT_REPO_ROOT/scripts/time_workload.py > $OUT/coverage.log 2>&1
cp $(find /tmp/$TAG/cov -name "*kernel_stats.csv" | head -1) $OUT/coverage_kernel_stats.csv; tail -6 $OUT/coverage.log
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_filter.py --out $OUT/bench_filter.json > $OUT/bench_filter.log 2>&1; tail -3 $OUT/bench_filter.log
TRACS_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --samples 2000 --sites 500000 --cpu-seconds 1 > $OUT/bench_n2.log 2>&1; grep '^{' $OUT/bench_n2.log | tail -1 > $OUT/bench_n2_gloo_one_gpu_2000x500000.json; cut -c1-300 $OUT/bench_n2_gloo_one_gpu_2000x500000.json
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras > $OUT/bench_c2.log 2>&1; tail -1 $OUT/bench_c2.log > $OUT/bench_c2.json; cut -c1-250 $OUT/bench_c2.json
timeout 900 python scripts/bench_e2e.py 10000 500000 > $OUT/e2e_10000x500000.json 2> $OUT/e2e.err; cut -c1-600 $OUT/e2e_10000x500000.json; tail -3 $OUT/e2e.err
timeout 900 python scripts/bench_e2e.py 2000 5000000 > $OUT/e2e_2000x5000000.json 2>> $OUT/e2e.err; cut -c1-300 $OUT/e2e_2000x5000000.json
timeout 300 python scripts/bench_e2e.py 10 100000 > $OUT/e2e_10x100000.json 2>> $OUT/e2e.err; cut -c1-300 $OUT/e2e_10x100000.json
timeout 900 python bench.py --partial 0.005 --steps 5 --warmup 2 --no-extras --cpu-seconds 1 > $OUT/bench_partial.log 2>&1; tail -1 $OUT/bench_partial.log > $OUT/bench_partial.json; cut -c1-250 $OUT/bench_partial.json
# the partial-code alignment's kernel trace (the table of profiles/r06/partial_floor.txt)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/partial -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --partial 0.005 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/partial_trace.log 2>&1
cp $(find /tmp/$TAG/partial -name "*kernel_stats.csv" | head -1) $OUT/partial_kernel_stats.csv
cd $GRAFT_REPO_ROOT
# the e2e command line with --filter
timeout 900 python scripts/bench_e2e.py 10000 500000 --filter > $OUT/e2e_10000x500000_filter.json 2>> $OUT/e2e.err; cut -c1-300 $OUT/e2e_10000x500000_filter.json
