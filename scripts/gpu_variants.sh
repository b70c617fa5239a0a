#!/bin/bash
# compile-time variants on one workload: usage (GPU box): WORKLOAD=sparse PARTIAL=0.005 bash scripts/gpu_variants.sh "-DA=1" "-DB=2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for F in "$@"; do
  echo "=== flags: $F"
  TRACS_EXTRA_HIPCC_FLAGS="$F" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  python scripts/time_workload.py 2>&1 | grep -E "per call|stages|kernels ms"
done
