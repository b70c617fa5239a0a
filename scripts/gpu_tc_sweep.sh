#!/bin/bash
# tc_long_keys_kernel: terms per lane and step x register budget (waves per SIMD); rebuilds the library on the box per variant
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "1 2" "2 3" "4 2" "4 3" "4 4" "2 4" "8 2"; do
  set -- $v
  TRACS_EXTRA_HIPCC_FLAGS="-DTRACS_TC_TPL=$1 -DTRACS_TC_WAVES=$2" python -m tracs_amd.build --force > /dev/null 2>&1
  echo "terms per lane $1, waves per SIMD $2:"; timeout 300 python scripts/probe_transcluster.py 2>&1 | grep "mean d"
done | tee gpurun_out/tc_sweep.txt
