#!/bin/bash
# compile-time variants over several workloads: usage (GPU box): bash scripts/gpu_variants_multi.sh "-DA=1" "-DB=2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for F in "$@"; do
  echo "=== flags: $F"
  TRACS_EXTRA_HIPCC_FLAGS="$F" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  for W in "sparse 0" "sparse 0.005" "coverage 0" "gappy 0" "lineage 0"; do
    set -- $W
    WORKLOAD=$1 PARTIAL=$2 timeout 120 python scripts/time_workload.py 2>&1 | grep -E "per call|stages" | cut -c1-230
  done
done
