cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /tmp/c2.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/c2/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if "tracs::" in r["Name"] and int(r["Calls"])>=25:
        per=float(r["TotalDurationNs"])/1e3/ (int(r["Calls"]))
        print("%-86s calls %4s avg %8.1f us" % (r["Name"][:86], r["Calls"], per))
PY
tail -1 /tmp/c2.log | cut -c1-200
