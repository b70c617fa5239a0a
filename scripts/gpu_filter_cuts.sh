#!/bin/bash
# Phase cuts / build variants of the filter's pair kernel (diagnostics; a cut's results are wrong by construction, only its time
# is read).  usage (GPU box): bash scripts/gpu_filter_cuts.sh "<flags of variant 1>" "<flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  python scripts/time_filter_kernel.py 2>&1 | tail -1
done
