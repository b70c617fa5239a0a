#!/bin/bash
# the minority budget (sites go to the lists while k (cN + k) <= n^2 / DIV) across the sensitivity workloads
cd "$GRAFT_REPO_ROOT" || exit 1
for DIV in "$@"; do
  echo "=== n^2 / $DIV"
  TRACS_MINOR_BUDGET_DIV=$DIV python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/budget_$DIV.json 2> gpurun_out/budget_$DIV.err || tail -5 gpurun_out/budget_$DIV.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/budget_$DIV.json").read().strip().splitlines()[-1])
print("  bench call %.2f ms" % d["ms_per_step"])
for k, w in d["sensitivity"]["workloads"].items():
    print("  %-10s first call %.2f ms  steady %.2f ms  kernels %s classes %s" % (k, w["single_pass_ms"], w["ms_per_pass"], {a: round(b, 2) for a, b in (w["kernels_ms"] or {}).items()}, w["site_classes"]))
PY
done
