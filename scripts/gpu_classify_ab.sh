cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for T in 256 128; do
  echo "== headline, TRACS_CLASSIFY_THREADS=$T"; TRACS_CLASSIFY_THREADS=$T WORKLOAD=sparse python scripts/time_workload.py 2>&1 | grep -E "per call|stages" | cut -c1-200
done; done
for T in 256 128; do
  echo "== partial, TRACS_CLASSIFY_THREADS=$T"; TRACS_CLASSIFY_THREADS=$T WORKLOAD=sparse PARTIAL=0.005 python scripts/time_workload.py 2>&1 | grep -E "per call|stages" | cut -c1-200
  echo "== coverage, TRACS_CLASSIFY_THREADS=$T"; TRACS_CLASSIFY_THREADS=$T WORKLOAD=coverage python scripts/time_workload.py 2>&1 | grep -E "per call|stages" | cut -c1-200
done
