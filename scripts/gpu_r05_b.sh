#!/bin/bash
# round 5, step b: site-class tests with the in-place cost model, config 2 in full, bench at config 2 and config 3 (short)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_scale.py -x -q -m gpu -k "not multirank and not rccl and not distance_cli and not two_ranks and not bench_line" > gpurun_out/r05b/tests.log 2>&1; tail -4 gpurun_out/r05b/tests.log
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras --cpu-seconds 1 > gpurun_out/r05b/bench_c2.log 2>&1; tail -1 gpurun_out/r05b/bench_c2.log > gpurun_out/r05b/bench_c2.json
timeout 900 python bench.py --steps 10 --warmup 2 --no-extras --cpu-seconds 1 > gpurun_out/r05b/bench_c3.log 2>&1; tail -1 gpurun_out/r05b/bench_c3.log > gpurun_out/r05b/bench_c3.json
python3 - <<PY
import json
for f in ("bench_c2", "bench_c3"):
    try:
        j = json.loads(open("gpurun_out/r05b/%s.json" % f).read())
        print(f, "ms/call %.3f steady %.3f tc %.3f" % (j["ms_per_step"], j["ms_per_step_steady_state"], j["config"]["transcluster_ms_per_step"]),
              j["roofline"].get("kernels_ms"), {s["stage"]: s["ms"] for s in j["roofline_per_pack"]["stages"]}, j["roofline"].get("site_classes", {}).get("counting_pass_in_place"))
    except Exception as e:
        print(f, "failed", e)
PY
