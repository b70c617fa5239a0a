#!/bin/bash
# round 4: row-structured transcluster passes (bounds, marking, gather) -- transcluster tests, the single-call probe, kernel trace
TAG=${1:-r04l}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_streams.py -q -m gpu -k "trans or stream or llk" > $OUT/t_tc.log 2>&1; tail -4 $OUT/t_tc.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "multirank or cli or rccl or config5" > $OUT/t_scale.log 2>&1; tail -4 $OUT/t_scale.log
timeout 600 python scripts/probe_single_pass.py 10000 5000000 once > $OUT/probe.log 2>&1; cat $OUT/probe.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- python3 $GRAFT_REPO_ROOT/scripts/probe_single_pass.py 10000 5000000 once > $OUT/trace.log 2>&1
cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/probe_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/probe_kernel_stats.csv")):
    if "tracs::" in r["Name"] and float(r["AverageNs"]) > 20000:
        print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
