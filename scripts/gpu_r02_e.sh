#!/bin/bash
# round 2: site classes -- tests, kernel variants, bench A/B (classes off / classes without minority lists / default)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for mode in "TRACS_SITE_CLASSES=0" "TRACS_MINORITY=0" "TRACS_NONE=1"; do
  echo "== $mode"
  env $mode timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_$mode.log 2>&1
  tail -1 $OUT/bench_$mode.log > $OUT/bench_$mode.json
  python3 - <<PY
import json
try:
    j=json.load(open("$OUT/bench_$mode.json")); r=j["roofline"]
    print(j["value"], j["ms_per_step"], j["config"]["checksum_d"], j["config"]["transcluster_ms_per_step"])
    print({k:r.get(k) for k in ("kernel","kernel_ms","frac","other_matrix_core_kernel","minority_lists_ms","dense_call_ms","site_classes")})
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench_$mode.log").read()[-2000:])
PY
done
