"""Thresholded (early-out) pair kernel vs the full pass on lineage-structured data.  usage: bench_thr.py [n] [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
d = torch.device("cuda", 0)
torch.manual_seed(3)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=d)
anc = torch.randint(0, 4, (L,), device=d, dtype=torch.uint8)
n_lin = 32
aln = dev.Alignment(n, L)
per = n // n_lin
for li in range(n_lin):                                   # samples grouped by lineage: ~2e-3*L SNPs between lineages, ~2e-5*L within
    f = anc.clone()
    m = torch.rand(L, device=d) < 1e-3
    f[m] = (f[m] + 1) & 3
    rows = f.unsqueeze(0).repeat(per, 1)
    mm = torch.rand((per, L), device=d) < 1e-5
    rows[mm] = (rows[mm] + 1) & 3
    aln.pack(lut[rows.long()], first=li * per)
dm = torch.zeros((n, n), dtype=torch.int32, device=d)
nm = torch.zeros((n, n), dtype=torch.int32, device=d)


def t(thr):
    dev.pairsnp_dense(aln, dm, nm, dist_threshold=thr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dev.pairsnp_dense(aln, dm, nm, dist_threshold=thr)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


full = t(None)
out = {"n": n, "L": L, "lineages": n_lin, "full_ms": full}
for thr in (100, 20):
    ms = t(thr)
    kept = int(((dm.view(torch.int32) >= 0) & (dm <= thr) & (torch.triu(torch.ones_like(dm), 1) > 0)).sum().item())
    out["thr_%d" % thr] = {"ms": ms, "speedup": full / ms, "pairs_emitted": kept}
print(json.dumps(out))
