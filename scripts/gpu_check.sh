#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_check.sh [pytest args...]'  -- the -m gpu suite (or a subset), log merged back under gpurun_out/check/
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/check
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- tests; fi
timeout 3300 python -m pytest "$@" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
