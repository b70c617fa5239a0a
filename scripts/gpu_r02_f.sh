#!/bin/bash
# round 2, site classes: full GPU suite, bench lines, rocprofv3 kernel trace of the same command, PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=30 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -A34 "slowest" $OUT/pytest.log | head -40; tail -3 $OUT/pytest.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json; cut -c1-300 $OUT/bench_c3.json
TRACS_SITE_CLASSES=0 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_c3_whole.json
TRACS_MINORITY=0 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_c3_nolists.json
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 10 --warmup 2 --no-extras 2>&1 | tail -1 > $OUT/bench_c2.json
timeout 900 python scripts/bench_distance_cli.py > $OUT/bench_distance_cli.json 2> $OUT/bench_distance_cli.log
TRACS_CLASSES_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "site classes" > $OUT/classes_trace.txt
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
rm -f $OUT/trace/trace_kernel_trace.csv
head -12 $OUT/trace/trace_kernel_stats.csv | cut -c1-200
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py 10000 400000 1"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-include-regex "pairsnp_|general_fixup" --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $T > $OUT/$name.log 2>&1; }
run m1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run m2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run l2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py 10000 5000000 1"
for c in FETCH_SIZE WRITE_SIZE; do
  run t_$c $c
done
python3 - <<PY > $OUT/pmc_r02.txt
import csv,collections
for d in ['m1','m2','l2','t_FETCH_SIZE','t_WRITE_SIZE']:
    try:
        rows=list(csv.DictReader(open('$OUT/'+d+'/pmc_counter_collection.csv')))
    except Exception as e:
        print(d,'missing',e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d, k, {c:"%.5g"%(sum(x)/len(x)) for c,x in v.items()})
PY
cat $OUT/pmc_r02.txt
for d in m1 m2 l2 t_FETCH_SIZE t_WRITE_SIZE; do rm -rf $OUT/$d; done
