#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -2
done
TRACS_EXTRA_HIPCC_FLAGS="" python -m tracs_amd.build --force > /dev/null 2>&1
PARTIAL=0.005 timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -2
timeout 900 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py -x -q -m gpu 2>&1 | tail -2
