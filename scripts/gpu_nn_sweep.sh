#!/bin/bash
# nn_rows_kernel: rounds (of four lists) in flight per wave x threads per workgroup (rebuilds the library on the box per variant)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "1 1024" "2 1024" "4 1024" "2 512" "4 512"; do
  set -- $v
  TRACS_EXTRA_HIPCC_FLAGS="-DTRACS_NN_FLIGHT=$1 -DTRACS_NN_THREADS=$2" python -m tracs_amd.build --force > /dev/null 2>&1
  echo "rounds in flight $1 threads $2: $(timeout 300 python scripts/probe_single_pass.py 2>&1 | grep -E "kernels" | tail -1)"
done | tee gpurun_out/nn_rows_sweep.txt
