#!/bin/bash
# round 5, step d: p lists split by w -- the lists' tests, the partial-code alignment at full size against the oracle, its call time and kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py -x -q -m gpu > gpurun_out/r05d/tests.log 2>&1; tail -3 gpurun_out/r05d/tests.log
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "full_size and partial" > gpurun_out/r05d/tests_full.log 2>&1; tail -3 gpurun_out/r05d/tests_full.log
timeout 900 python bench.py --partial 0.005 --steps 5 --warmup 2 --no-extras --cpu-seconds 1 > gpurun_out/r05d/bench_partial.log 2>&1; tail -1 gpurun_out/r05d/bench_partial.log > gpurun_out/r05d/bench_partial.json
python3 - <<PY
import json
j = json.loads(open("gpurun_out/r05d/bench_partial.json").read())
print("partial: ms/call %.3f steady %.3f tc %.3f" % (j["ms_per_step"], j["ms_per_step_steady_state"], j["config"]["transcluster_ms_per_step"]), j["roofline"].get("kernels_ms"), {s["stage"]: s["ms"] for s in j["roofline_per_pack"]["stages"]})
PY
bash scripts/gpu_prof_cmd.sh r05d_partial $GRAFT_REPO_ROOT/bench.py --partial 0.005 --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r05d_partial_kernel_stats.csv")):
    if "tracs::" in r["Name"] and float(r["AverageNs"]) > 50000:
        print("    %-60s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
timeout 600 python scripts/bench_e2e.py 10000 500000 > gpurun_out/r05d/e2e_10000x500000.json 2> gpurun_out/r05d/e2e.err
python3 -c "
import json; j=json.load(open('gpurun_out/r05d/e2e_10000x500000.json')); print('e2e', j['command_seconds'], j['command_seconds_first_run_on_the_box']); [print('   ', s['stage'][:80], s['seconds']) for s in j['stages']]"
