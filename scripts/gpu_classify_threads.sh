cd "$GRAFT_REPO_ROOT" || exit 1
echo "== tests with 128 threads forced"; TRACS_CLASSIFY_THREADS=128 timeout 1200 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_golden.py tests/test_gpu_lists.py -x -q 2>&1 | tail -3
TRACS_CLASSIFY_THREADS=128 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "full_size and not filter" 2>&1 | tail -2
echo "== tests, default choice"; timeout 1200 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3
for T in 256 128; do
  echo "== config 2, TRACS_CLASSIFY_THREADS=$T"
  TRACS_CLASSIFY_THREADS=$T python bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], [(s['stage'], s['ms']) for s in j['roofline_per_pack']['stages']])"
done
for T in 256 128; do
  echo "== headline, TRACS_CLASSIFY_THREADS=$T"; TRACS_CLASSIFY_THREADS=$T WORKLOAD=sparse python scripts/time_workload.py 2>&1 | grep -E "per call|stages"
done
