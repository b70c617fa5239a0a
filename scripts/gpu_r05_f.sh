#!/bin/bash
# round 5, step f: the rows' bitmaps written by the per-site pass (TRACS_FUSE_BITMAPS=1, the default) against n_bitmap_kernel (=0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05f
for F in 1 0; do
  echo "=== TRACS_FUSE_BITMAPS=$F"
  TRACS_FUSE_BITMAPS=$F timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels|warm:" | tail -3
done
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py -x -q -m gpu > gpurun_out/r05f/tests.log 2>&1; tail -3 gpurun_out/r05f/tests.log
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_scale.py -x -q -m gpu -k "full_size or config2 or multirank" > gpurun_out/r05f/tests_full.log 2>&1; tail -3 gpurun_out/r05f/tests_full.log
