cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_site_classes.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
for E in 0 1; do
  for P in 0.005 0; do
    echo "=== TRACS_CLASSIFY_EMIT=$E partial=$P"
    TRACS_CLASSIFY_EMIT=$E WORKLOAD=sparse PARTIAL=$P python scripts/time_workload.py 2>&1 | grep -E "per call|stages|kernels ms"
  done
done
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "full_size and not filter" 2>&1 | tail -3
