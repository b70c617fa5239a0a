"""Many samples through the dense path: 50 000 samples (the sample count of BASELINE config 4) x 100 kbp generated on the
device, all 1.25e9 pairs on one GPU, d / nn checked against numpy on sampled blocks (first, middle, last samples and their
cross blocks).  Guards tile counts, n_pad and 64-bit cell offsets at large n.  usage: python scripts/check_large_n.py [n] [L]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402
from tracs_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
keep_ranges = [(0, 48), (n // 2 - 7, n // 2 + 41), (n - 48, n)]
kept = {}


def emit(rows, first):
    aln.pack(rows, first=first)
    for a, b in keep_ranges:
        lo, hi = max(a, first), min(b, first + rows.shape[0])
        if lo < hi:
            kept.update({s: rows[s - first].cpu().numpy() for s in range(lo, hi)})


t0 = time.perf_counter()
aln = dev.Alignment(n, L)
synth.generate_device(n, L, 99, emit, mu_lineage=2e-3, mu_sample=2e-4, n_lineages=40, p_n=0.01, batch=256)
out = {"samples": n, "sites": L, "pairs": n * (n - 1) // 2, "setup_s": time.perf_counter() - t0}
d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
dev.pairsnp_dense(aln, d, nn)
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.pairsnp_dense(aln, d, nn)
torch.cuda.synchronize()
out["dense_s"] = time.perf_counter() - t0
out["pairs_per_s"] = out["pairs"] / out["dense_s"]
out["encoding"], out["kernel"] = aln.encoding, aln.kernel
ids = sorted(kept)
seqs = np.array([kept[s] for s in ids])
valid = np.isin(seqs, np.frombuffer(b"ACGT", np.uint8))
idx = torch.tensor(ids, device="cuda")
gd = d[idx][:, idx].cpu().numpy()
gn = nn[idx][:, idx].cpu().numpy()
checked = 0
for a in range(len(ids)):
    both = valid[a][None, :] & valid
    dd = ((seqs[a][None, :] != seqs) & both).sum(1)
    cc = both.sum(1)
    for b in range(a + 1, len(ids)):
        assert gd[a, b] == dd[b] and gn[a, b] == cc[b], (ids[a], ids[b], int(gd[a, b]), int(dd[b]), int(gn[a, b]), int(cc[b]))
        checked += 1
out["pairs_checked"] = checked
# thresholded two-pass run: every pair <= threshold exact, COO count equals the dense count
thr = 60
d2 = torch.zeros_like(d)
dev.pairsnp_dense(aln, d2, None, dist_threshold=thr)
close = torch.triu((d <= thr), 1)
assert not bool(((d2 != d) & close).any().item())
rows, cols, dd, _ = dev.coo_from_dense(d2, None, n, dist_threshold=thr)
assert rows.numel() == int(close.sum(dtype=torch.int64).item())
out["pairs_within_threshold"] = int(rows.numel())
print(json.dumps(out))
