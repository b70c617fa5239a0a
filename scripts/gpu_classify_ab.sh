cd "$GRAFT_REPO_ROOT" || exit 1
echo "== tests with 64 threads forced"; TRACS_CLASSIFY_THREADS=64 timeout 1200 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_golden.py -x -q 2>&1 | tail -2
for rep in 1 2; do for T in 128 64; do
  echo "== headline, TRACS_CLASSIFY_THREADS=$T"; TRACS_CLASSIFY_THREADS=$T WORKLOAD=sparse python scripts/time_workload.py 2>&1 | grep -E "per call|stages" | cut -c1-200
done; done
for T in 128 64; do
  echo "== config 2, TRACS_CLASSIFY_THREADS=$T"
  TRACS_CLASSIFY_THREADS=$T python bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], [(s['stage'], s['ms']) for s in j['roofline_per_pack']['stages']])"
done
