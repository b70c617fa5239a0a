#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_configs.py tests/test_filter_recomb.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp
TRACS_BENCH_PARTIAL=0.005 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_partial -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace_partial.log 2>&1
rm -f $OUT/trace_partial/trace_kernel_trace.csv
grep -E "pairsnp_mfma|general_fixup" $OUT/trace_partial/trace_kernel_stats.csv | cut -c1-60,300-420
cd $GRAFT_REPO_ROOT
# thresholded general pass at full size: matrix cores vs VALU
python - <<'PY' 2>&1 | tee $OUT/thr_general.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tracs_amd import device as dev, synth
n, L = 10000, 5000000
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=20241024, mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=0.0005)
d = torch.zeros((n, n), dtype=torch.int32, device="cuda"); nn = torch.zeros_like(d)
dev.pairsnp_dense(aln, d, nn); torch.cuda.synchronize()
mean = float(d.double().sum().item()) / (n * (n - 1) / 2)
for thr in (None, int(mean) - 100, int(mean) - 300):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.pairsnp_dense(aln, d, nn, dist_threshold=thr); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    kept = int((torch.triu(d, 1) > 0).sum().item()) if thr is None else int(((torch.triu(d, 1) > 0) & (d <= thr)).sum().item())
    print("threshold", thr, "kernel", aln.kernel, "ms %.1f" % (dt * 1e3), "pairs within", kept, "mean d %.1f" % mean)
PY
