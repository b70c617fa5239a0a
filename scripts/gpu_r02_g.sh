#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02x
mkdir -p $OUT
export TMPDIR=/tmp
for shape in 4x2 4x2g2 4x2g8 2x2g2 2x2g8 2x4; do
  TRACS_COUNT_TILE=$shape timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/b_$shape.json
  python3 - <<PY
import json
j=json.load(open("$OUT/b_$shape.json")); r=j["roofline"]
print("$shape", round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), j["config"]["checksum_d"])
PY
done
