#!/bin/bash
# round 4: list kernels -- parity subset, probe, rocprofv3 kernel trace + PMC passes of the list kernels (10 000 x 1 Mbp: same rows, a fifth of the sites)
TAG=${1:-r04c}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py -x -q -m gpu > $OUT/t_lists.log 2>&1; tail -5 $OUT/t_lists.log
timeout 1500 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -q -m gpu > $OUT/t_classes.log 2>&1; tail -15 $OUT/t_classes.log
timeout 600 python scripts/probe_single_pass.py > $OUT/probe.log 2>&1; cat $OUT/probe.log
cd /tmp
P="python3 $GRAFT_REPO_ROOT/scripts/probe_single_pass.py 10000 1000000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- $P > $OUT/trace.log 2>&1
cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/probe_1M_kernel_stats.csv; grep -E "tracs::" $OUT/probe_1M_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160 | head -30
RX="nn_rows_kernel|site_lists_kernel|n_bitmap_kernel|minor_fixup_kernel|classify_sites"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-include-regex "$RX" --pmc $C --output-format csv -d /tmp/$TAG/pmc_$i -o pmc -- $P > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY > $OUT/pmc_lists.txt
import csv, collections, glob
for i in range(1, 6):
    fs = glob.glob('/tmp/$TAG/pmc_%d/**/*counter_collection.csv' % i, recursive=True)
    if not fs:
        print(i, 'missing'); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(i, '|', k, '|', {c: ("%.6g" % (sum(x) / len(x)), len(x)) for c, x in v.items()})
PY
cat $OUT/pmc_lists.txt
