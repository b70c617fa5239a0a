cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 $GRAFT_REPO_ROOT/bench.py --partial 0.005 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/pp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tracs::" in r["Name"] and float(r["AverageNs"])>8e4:
        print("%-86s calls %4s avg %8.1f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"])/1e3))
PY
tail -1 /tmp/pp.log | cut -c1-160
