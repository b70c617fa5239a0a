#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ac
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
j=json.load(open("$OUT/bench.json")); r=j["roofline"]
print(round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), r["minority_lists_ms"], j["config"]["checksum_d"], j["config"]["transcluster_ms_per_step"], j["config"]["distinct_keys"])
PY
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
