#!/bin/bash
# round 4: two-phase list walk, 128-thread per-site pass -- list / class tests, probe, bench line
TAG=${1:-r04f}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py -q -m gpu > $OUT/t_lists.log 2>&1; tail -4 $OUT/t_lists.log
timeout 1500 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -q -m gpu > $OUT/t_classes.log 2>&1; tail -12 $OUT/t_classes.log
timeout 600 python scripts/probe_single_pass.py > $OUT/probe.log 2>&1; cat $OUT/probe.log
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
