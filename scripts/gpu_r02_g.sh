#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/check_large_n.py > $OUT/check_large_n.json 2> $OUT/check_large_n.log; echo "rc $?"; cat $OUT/check_large_n.json | cut -c1-600; tail -3 $OUT/check_large_n.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
j=json.load(open("$OUT/bench.json")); r=j["roofline"]
print(round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), r["minority_lists_ms"], j["config"]["checksum_d"], j["config"]["transcluster_ms_per_step"], j["config"]["per_pack_decisions_ms"])
g=j["roofline_general"]; print("general", {k: g.get(k) for k in ("kernel","kernel_ms","frac","other_matrix_core_kernel","lists_ms","dense_call_ms","site_classes","mean_d")})
PY
