#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ag
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
j=json.load(open("$OUT/bench.json")); r=j["roofline"]
print(round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), r["minority_lists_ms"], j["config"]["checksum_d"], j["config"]["transcluster_ms_per_step"])
g=j["roofline_general"]; print("general", {k:g.get(k) for k in ("kernel_ms","frac","lists_ms","dense_call_ms","mean_d")})
PY
timeout 1800 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_configs.py tests/test_gpu_random.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
