"""Warm time of tracs_filter_recomb_pairs over every pair of the config-3 alignment (diagnostics: phase cuts and kernel variants).
No result check -- scripts/bench_filter.py and tests/test_filter_recomb.py do that."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench as B  # noqa: E402
from tracs_amd import device as dev, synth  # noqa: E402

n, L = int(os.environ.get("N", "10000")), int(os.environ.get("SITES", "5000000"))
wl, partial = os.environ.get("WORKLOAD", "sparse"), float(os.environ.get("PARTIAL", "0"))
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=20241022 + 2, **B.synth_kw(partial, wl))
dmat = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nmat = torch.zeros((n, n), dtype=torch.int32, device="cuda")
dev.pairsnp_dense(aln, dmat, nmat)
rows, cols, d, _ = dev.coo_from_dense(dmat, nmat, n)
del nmat
try:
    dev.filter_recomb_pairs(aln, rows, cols, d)
except RuntimeError as e:          # a cut build fails the consistency check by construction
    print("(first call: %s)" % e)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    try:
        dev.filter_recomb_pairs(aln, rows, cols, d)
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t)
print("index", dev.filter_index_info(aln))
print("filter over %d pairs: %.1f ms (%s, partial %g)" % (rows.numel(), best * 1e3, wl, partial))
