cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/rand_lines.hip -o /tmp/rand_lines 2>/dev/null
cd /tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/rl_pmc
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/rl_pmc -o pmc -- /tmp/rand_lines 256 > /tmp/rl.log 2>&1
  python3 - <<PY
import csv, glob, collections
fs = glob.glob('/tmp/rl_pmc/**/*counter_collection.csv', recursive=True)
if not fs:
    print("$c: no counters (", open('/tmp/rl.log').read()[-300:], ")")
else:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        agg.setdefault((r['Kernel_Name'][:40], r['Counter_Name']), []).append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k, ["%.4g" % x for x in v])
PY
done
