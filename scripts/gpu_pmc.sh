#!/bin/bash
# usage: bash scripts/gpu_pmc.sh <tag> <samples> <sites>  -- PMC passes restricted to the pair kernel
TAG=$1; NS=$2; NL=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
T="python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $NS $NL 2"
python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $NS $NL 3 2>&1 | tail -1 | tee $OUT/plain.txt
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-include-regex pairsnp_tile --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $T > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log | cut -c1-200; }
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $T > $OUT/trace.log 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run sq3 SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_WAVES_EQ_64
run l2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
find $OUT -name "*.csv" -size +4M -delete
ls $OUT/*
