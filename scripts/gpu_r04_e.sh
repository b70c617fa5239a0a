#!/bin/bash
# round 4: per-site pass by bit transposition, CSA classification, site-sharded CLI -- tests, probe, PMC of the list walk, bench line
TAG=${1:-r04e}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py -q -m gpu > $OUT/t_lists.log 2>&1; tail -4 $OUT/t_lists.log
timeout 1500 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_random.py -q -m gpu > $OUT/t_classes.log 2>&1; tail -12 $OUT/t_classes.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "rccl or multirank or two_ranks" > $OUT/t_scale.log 2>&1; tail -12 $OUT/t_scale.log
timeout 600 python scripts/probe_single_pass.py > $OUT/probe.log 2>&1; cat $OUT/probe.log
cd /tmp
P="python3 $GRAFT_REPO_ROOT/scripts/probe_single_pass.py 10000 1000000 once"
RX="nn_rows_kernel|site_lists_kernel|classify_sites"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-include-regex "$RX" --pmc $C --output-format csv -d /tmp/$TAG/pmc_$i -o pmc -- $P > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY > $OUT/pmc_lists.txt
import csv, collections, glob
for i in range(1, 5):
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
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log > $OUT/bench.json; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for k in ("value", "ms_per_step", "value_steady_state", "ms_per_step_steady_state", "value_worst_workload"):
    print(k, d.get(k))
print("tc ms", d["config"].get("transcluster_ms_per_step"))
for st in d["roofline_per_pack"]["stages"]:
    print(st)
print({k: v for k, v in d["roofline"].items() if k in ("kernel", "kernel_ms", "frac", "frac_needed", "frac_lines", "kernels_ms")})
for w, r in d.get("sensitivity", {}).get("workloads", {}).items():
    print(w, round(r["ms_per_pass"], 2), round(r["single_pass_ms"], 1), r["site_classes"], r["kernels_ms"])
print("general", {k: d["roofline_general"].get(k) for k in ("dense_call_ms", "kernels_ms")})
PY
