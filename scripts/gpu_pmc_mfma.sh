#!/bin/bash
# usage: bash scripts/gpu_pmc_mfma.sh <tag> <samples> <sites>  -- matrix-core kernel: MFMA busy cycles, instruction counts, waits, L2
TAG=$1; NS=$2; NL=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $NS $NL 1"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-include-regex "pairsnp_" --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $T > $OUT/$name.log 2>&1; }
run m1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run m2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run l2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv,collections
for d in ['m1','m2','l2']:
    try:
        rows=list(csv.DictReader(open('$OUT/'+d+'/pmc_counter_collection.csv')))
    except Exception as e:
        print(d,'missing',e); continue
    agg=collections.defaultdict(list)
    for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
