"""filter_recomb throughput: emitted pairs/s of the extract + test kernels at a given alignment length.
usage: python scripts/bench_filter.py [samples] [sites]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402
from tracs_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=9, mu_lineage=2e-4, mu_sample=2e-5, n_lineages=16, p_n=0.01)
d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
dev.pairsnp_dense(aln, d, nn)
out = {"samples": n, "sites": L, "encoding": aln.encoding}
for thr in (100, 2147483647):
    rows, cols, dd, nc = dev.coo_from_dense(d, nn, n, dist_threshold=thr)
    if rows.numel() > 4000000:
        rows, cols, dd = rows[:4000000], cols[:4000000], dd[:4000000]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    filt, found, pos, off = dev.filter_recomb_device(aln, rows, cols, dd)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    t0 = time.perf_counter()
    filt, found, pos, off = dev.filter_recomb_device(aln, rows, cols, dd)
    torch.cuda.synchronize()
    t = min(t, time.perf_counter() - t0)
    out["thr_%d" % thr] = {"pairs": int(rows.numel()), "snps": int(dd.sum().item()), "s": t, "pairs_per_s": rows.numel() / t,
                           "plane_GBps": rows.numel() * L * (0.75 if aln.encoding == "consensus" else 1.0) / t / 1e9,
                           "removed": int((dd - filt).sum().item())}
print(json.dumps(out))
