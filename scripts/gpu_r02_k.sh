#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_quick.json
python - <<PY
import json
d=json.load(open('$OUT/bench_quick.json'))
print({k:d[k] for k in ('value','ms_per_step')}, 'tc ms', d['config']['transcluster_ms_per_step'], 'kernel ms', d['roofline']['kernel_ms'])
PY
