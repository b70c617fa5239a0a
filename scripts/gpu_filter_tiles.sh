#!/bin/bash
# tile size of flt_pairs_tiled_kernel (diagnostics): usage (GPU box): bash scripts/gpu_filter_tiles.sh 512 1024 2048
cd "$GRAFT_REPO_ROOT" || exit 1
for T in "$@"; do
  echo "=== tile $T $FLAGS"
  TRACS_EXTRA_HIPCC_FLAGS="-DTRACS_FLT_TILE=$T $FLAGS" python -m tracs_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  python scripts/bench_filter.py --partial 0.005 --check 8 --scan-sample 100000 2>&1 | grep -E "warm_call_s|equal"
done
