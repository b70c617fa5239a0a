#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for shape in 4x4 4x2 2x2; do
  TRACS_COUNT_TILE=$shape timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/b_$shape.json
  python3 - <<PY
import json
j=json.load(open("$OUT/b_$shape.json")); r=j["roofline"]
print("$shape", round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), r.get("minority_lists_ms"), j["config"].get("per_pack_decisions_ms"))
PY
done
