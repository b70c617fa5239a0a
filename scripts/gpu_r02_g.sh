#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ae
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
python3 - <<PY
import json
j=json.load(open("$OUT/bench.json")); r=j["roofline"]
print(round(j["ms_per_step"],2), r["kernel"], round(r["kernel_ms"],2), round(r["frac"],4), r["minority_lists_ms"], j["config"]["checksum_d"], j["config"]["transcluster_ms_per_step"], j["config"]["distinct_keys"])
PY
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c2', j['ms_per_step'], j['config']['transcluster_ms_per_step'])"
timeout 2400 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_oracle_vs_ref.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
