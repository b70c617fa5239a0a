#!/bin/bash
# usage: bash scripts/gpu_pmc_kernel.sh <tag> <kernel regex> <python script and args...>
# separate rocprofv3 --pmc passes (FETCH/WRITE, L2, SQ wait / LDS counters) of the kernels matching the regex; summary on stdout
TAG=$1; RE=$2; shift; shift
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-include-regex "$RE" --pmc "$@" --output-format csv -d /tmp/pmc_$TAG/$name -o pmc -- python3 $CMD > /tmp/pmc_$TAG.$name.log 2>&1; }
CMD="$*"
run f FETCH_SIZE
run w WRITE_SIZE
run l2 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
run s1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
run s2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
python3 - <<PY | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
import csv, collections, glob
for d in ['f', 'w', 'l2', 's1', 's2']:
    fs = glob.glob('/tmp/pmc_$TAG/' + d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(d, 'missing'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, '|', k, '|', {c: "%.5g" % (sum(x) / len(x)) for c, x in v.items()})
PY
