#!/bin/bash
# round 4: the whole GPU suite, as the driver runs it (without -x: every failure listed)
TAG=${1:-r04tests}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3300 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1; tail -40 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
