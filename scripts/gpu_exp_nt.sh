#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "" "-DTRACS_EXP_NT" ; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels|repeat" | tail -5
done
