#!/bin/bash
# probe_single_pass.py (stages + kernels) for several builds: bash scripts/gpu_exp_nn.sh "<flags 1>" "<flags 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -3
done
