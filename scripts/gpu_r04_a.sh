#!/bin/bash
# round 4, first GPU call: the new bench line (one call per step), stream API tests, classify without the N plane
TAG=${1:-r04a}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_site_classes.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for k in ("value", "ms_per_step", "value_steady_state", "ms_per_step_steady_state"):
    print(k, d.get(k))
print(json.dumps(d.get("roofline_per_pack"), indent=1)[:3000])
print(json.dumps(d.get("single_pass"), indent=1)[:1500])
PY
