#!/bin/bash
# nn_rows_kernel (bench alignment) for several builds: usage (GPU box): bash scripts/gpu_exp_nn.sh "<flags>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  for r in 1 2; do timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "kernels" | tail -2 | cut -c1-120; done
done
