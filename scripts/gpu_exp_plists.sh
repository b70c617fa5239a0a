#!/bin/bash
# p_lists_kernel on the partial-code alignment for several builds: usage (GPU box): bash scripts/gpu_exp_plists.sh "<flags>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for V in "$@"; do
  echo "=== $V"
  TRACS_EXTRA_HIPCC_FLAGS="$V" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  PARTIAL=0.005 bash scripts/gpu_prof_cmd.sh plx $GRAFT_REPO_ROOT/scripts/probe_single_pass.py > /dev/null 2>&1
  grep -E "p_lists|site_lists" gpurun_out/plx_kernel_stats.csv | sed 's/(.*)",/",/' | cut -c1-120
done
