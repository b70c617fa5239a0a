#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02af
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q -s -k "trans_dist" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "large-N|E\(K\) keys|passed|failed|rc" $OUT/pytest.log
