#!/bin/bash
# full GPU suite + one driver-style bench line; logs under gpurun_out/<tag>/
TAG=${1:-suite}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -A18 "slowest" $OUT/pytest.log | head -22; tail -4 $OUT/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-400 $OUT/bench.json
