#!/bin/bash
# per-site pass of the partial-code alignment with phases cut out (results wrong: timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "" "-DTRACS_EXP_NOATOM" ; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  PARTIAL=0.005 timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -2
done
