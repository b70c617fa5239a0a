#!/bin/bash
# usage: bash scripts/gpu_prof.sh <tag> [samples] [sites]  -- rocprofv3 kernel trace + PMC passes of bench.py
TAG=${1:-p01}; NS=${2:-10000}; NL=${3:-5000000}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --samples $NS --sites $NL --steps 2 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $BENCH > $OUT/pmc_l2.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
ls -R $OUT | head -50
# keep only small summaries (the per-dispatch CSVs are small here: a few hundred dispatches)
find $OUT -name "*.csv" -size +8M -delete
