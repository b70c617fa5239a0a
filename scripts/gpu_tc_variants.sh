#!/bin/bash
# transcluster alone (scripts/probe_transcluster.py, bench-like keys) for several builds of the term-ratio kernel:
# usage (GPU box): bash scripts/gpu_tc_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  N=10000 python scripts/probe_transcluster.py 1000 2>&1 | tail -2
  bash scripts/gpu_prof_cmd.sh tcv $GRAFT_REPO_ROOT/scripts/probe_transcluster.py 1000 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/tcv_kernel_stats.csv")):
    if "tracs::tc_" in r["Name"]:
        print("    %-44s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
