#!/bin/bash
# minor_fixup_kernel on the partial-code alignment with phases cut out (results wrong: timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  PARTIAL=0.005 timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "kernels" | tail -1
done
