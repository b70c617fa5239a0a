#!/bin/bash
# round 5, step c: the command line with the results on the device until the CSV rows -- golden CSVs through both paths, two-rank CLI,
# end to end at 10 000 x 500 kbp and 2 000 x 5 Mbp, config 1 (10 x 100 kb) wall time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05c
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "cli or parity or config1 or fixture or drop_in" > gpurun_out/r05c/tests.log 2>&1; tail -4 gpurun_out/r05c/tests.log
timeout 900 python scripts/bench_e2e.py 10000 500000 > gpurun_out/r05c/e2e_10000x500000.json 2> gpurun_out/r05c/e2e.err; tail -3 gpurun_out/r05c/e2e.err
timeout 900 python scripts/bench_e2e.py 2000 5000000 > gpurun_out/r05c/e2e_2000x5000000.json 2>> gpurun_out/r05c/e2e.err
timeout 300 python scripts/bench_e2e.py 10 100000 > gpurun_out/r05c/e2e_10x100000.json 2>> gpurun_out/r05c/e2e.err
python3 - <<PY
import json
for f in ("e2e_10000x500000", "e2e_2000x5000000", "e2e_10x100000"):
    try:
        j = json.loads(open("gpurun_out/r05c/%s.json" % f).read())
        print(f, "command %.3f s (first %.3f)" % (j["command_seconds"], j["command_seconds_first_run_on_the_box"]))
        for s in j["stages"]:
            print("    %-90s %.4f" % (s["stage"][:90], s["seconds"]))
    except Exception as e:
        print(f, "failed", e)
PY
