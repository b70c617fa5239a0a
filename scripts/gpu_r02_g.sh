#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $OUT
export TMPDIR=/tmp
TRACS_CLASSES_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
grep "site classes" $OUT/trace.log
tail -1 $OUT/trace.log | cut -c1-200
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-200 $OUT/bench.json
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
