#!/bin/bash
# round 4: list kernels after the first tuning pass -- tests, probe (+ timing variants of the per-site pass), bench line
TAG=${1:-r04d}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lists.py -q -m gpu > $OUT/t_lists.log 2>&1; tail -4 $OUT/t_lists.log
timeout 1500 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_gpu_golden.py -q -m gpu > $OUT/t_classes.log 2>&1; tail -12 $OUT/t_classes.log
timeout 600 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "rccl or multirank or two_ranks" > $OUT/t_scale.log 2>&1; tail -12 $OUT/t_scale.log
timeout 600 python scripts/probe_single_pass.py > $OUT/probe.log 2>&1; cat $OUT/probe.log
for v in 1 2 3; do
  TRACS_DBG_SITE=$v timeout 600 python scripts/probe_single_pass.py 10000 5000000 once > $OUT/probe_dbg$v.log 2>&1; echo "TRACS_DBG_SITE=$v (timing only)"; grep -E "stages" $OUT/probe_dbg$v.log | head -2
done
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log > $OUT/bench.json; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for k in ("value", "ms_per_step", "value_steady_state", "ms_per_step_steady_state", "value_worst_workload"):
    print(k, d.get(k))
print("tc ms", d["config"].get("transcluster_ms_per_step"))
for st in d["roofline_per_pack"]["stages"]:
    print(st)
print({k: v for k, v in d["roofline"].items() if k in ("kernel", "kernel_ms", "frac", "frac_needed", "frac_lines", "kernels_ms")})
for w, r in d.get("sensitivity", {}).get("workloads", {}).items():
    print(w, round(r["ms_per_pass"], 2), round(r["single_pass_ms"], 1), r["site_classes"], r["kernels_ms"])
PY
