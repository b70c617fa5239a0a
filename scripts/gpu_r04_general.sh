#!/bin/bash
# round 4: the same alignment with 0.5 % partial IUPAC codes (general encoding): which kernels a call spends its time in
TAG=${1:-r04gen}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --partial 0.005 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/bench_general.log 2>&1
tail -1 $OUT/bench_general.log | cut -c1-300
cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/general_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/general_kernel_stats.csv")):
    if "tracs::" in r["Name"] and float(r["AverageNs"]) > 50000 and "pack_kernel" not in r["Name"]:
        print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
