#!/bin/bash
# round 2: new config tests, full-size bench line (all extras), PMC counters of the matrix-core kernels
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_align_stage.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json; tail -c 3000 $OUT/bench_c3.json
cd /tmp
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py 10000 400000 1"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-include-regex "pairsnp_" --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $T > $OUT/$name.log 2>&1; }
run m1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run m2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
export TRACS_FORCE_GENERAL=1
run g1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run g2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
unset TRACS_FORCE_GENERAL
python3 - <<PY > $OUT/pmc_mfma_10000x400000.txt
import csv,collections
for d in ['m1','m2','g1','g2']:
    try:
        rows=list(csv.DictReader(open('$OUT/'+d+'/pmc_counter_collection.csv')))
    except Exception as e:
        print(d,'missing',e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d, k, {c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
cat $OUT/pmc_mfma_10000x400000.txt
rm -rf $OUT/m1 $OUT/m2 $OUT/g1 $OUT/g2
