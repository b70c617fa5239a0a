#!/bin/bash
# round 5, step e: q lines -- lists / site-class / variant tests, partial codes at full size vs the oracle, call times (partial, consensus)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r05e/tests.log 2>&1; tail -3 gpurun_out/r05e/tests.log
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "full_size and (partial or divergent)" > gpurun_out/r05e/tests_full.log 2>&1; tail -3 gpurun_out/r05e/tests_full.log
for W in "--partial 0.005" "" "--workload divergent"; do
timeout 900 python bench.py $W --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05e/bench.log 2>&1; tail -1 gpurun_out/r05e/bench.log > "gpurun_out/r05e/bench$(echo $W | tr -d ' -.').json"
python3 - <<PY
import json
j = json.loads(open("gpurun_out/r05e/bench.log").read().strip().splitlines()[-1])
print("$W: ms/call %.3f steady %.3f tc %.3f" % (j["ms_per_step"], j["ms_per_step_steady_state"], j["config"]["transcluster_ms_per_step"]), j["roofline"].get("kernels_ms"), {s["stage"]: s["ms"] for s in j["roofline_per_pack"]["stages"]})
PY
done
