#!/bin/bash
# round 5, step g: four-bytes-per-round n8 encoder (lists tests, call time), the CLI's device path in small batches, end to end again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_kernel_variants.py tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r05g/tests.log 2>&1; tail -3 gpurun_out/r05g/tests.log
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_scale.py -x -q -m gpu -k "cli or small_batches or config2" > gpurun_out/r05g/tests_cli.log 2>&1; tail -3 gpurun_out/r05g/tests_cli.log
timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels|warm:" | tail -3
PARTIAL=0.005 timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -2
timeout 600 python scripts/bench_e2e.py 10000 500000 > gpurun_out/r05g/e2e_10000x500000.json 2> gpurun_out/r05g/e2e.err
timeout 600 python scripts/bench_e2e.py 2000 5000000 > gpurun_out/r05g/e2e_2000x5000000.json 2>> gpurun_out/r05g/e2e.err
python3 - <<PY
import json
for f in ("e2e_10000x500000", "e2e_2000x5000000"):
    j = json.load(open("gpurun_out/r05g/%s.json" % f)); print(f, j["command_seconds"], j["command_seconds_first_run_on_the_box"])
    for s in j["stages"]: print("    %-85s %.4f" % (s["stage"][:85], s["seconds"]))
PY
