"""> 4 GB through the HOST path: FASTA on disk -> tracs_alignment_from_fasta -> dense pair kernel -> COO, checked against
numpy on sampled pairs (guards the size_t / 32-bit offset paths the small fixtures cannot reach).
usage: python scripts/check_large_fasta.py [samples] [sites]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import api  # noqa: E402
from tracs_amd import device as dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
tmp = os.environ.get("TMPDIR", "/tmp")
path = os.path.join(tmp, "large.fa")
rng = np.random.default_rng(77)
base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
alphabet = np.frombuffer(b"ACGTNRYKMSWacgtn-", np.uint8)
keep = {}
t0 = time.perf_counter()
with open(path, "wb") as f:
    for s in range(n):
        row = base.copy()
        pos = rng.integers(0, L, 400)
        row[pos] = alphabet[rng.integers(0, len(alphabet), 400)]
        if s % 97 == 0 or s >= n - 3:
            keep[s] = row.copy()
        f.write(b">s%d\n" % s)
        f.write(row.tobytes())
        f.write(b"\n")
out = {"samples": n, "sites": L, "file_GB": os.path.getsize(path) / 1e9, "write_s": time.perf_counter() - t0}


def mask(row):
    lut = np.full(256, 15, np.uint8)
    for ch, m in zip(b"ACGTMRWSYKVHDB", (1, 2, 4, 8, 3, 5, 9, 6, 10, 12, 7, 11, 13, 14)):
        lut[ch] = m
        lut[ch + 32] = m
    return lut[row]


t0 = time.perf_counter()
aln = dev.Alignment.from_fasta([path])
torch.cuda.synchronize()
out["ingest_s"] = time.perf_counter() - t0
assert aln.n == n and aln.L == L and aln.names[-1] == "s%d" % (n - 1)
d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
dev.pairsnp_dense(aln, d, nn)
torch.cuda.synchronize()
dh, nh = d.cpu().numpy(), nn.cpu().numpy()
ids = sorted(keep)
checked = 0
for a in range(len(ids)):
    for b in range(a + 1, len(ids)):
        i, j = ids[a], ids[b]
        mi, mj = mask(keep[i]), mask(keep[j])
        assert dh[i, j] == int(np.count_nonzero((mi & mj) == 0)), (i, j)
        assert nh[i, j] == int(np.count_nonzero((mi != 15) & (mj != 15))), (i, j)
        checked += 1
out["pairs_checked_dense"] = checked
aln.close()
del d, nn
# the reference-shaped host call on the same file, thresholded so the COO stays small
t0 = time.perf_counter()
r, c, dd, names, _, ncomp = api.pairsnp_arrays([path], 1, 700, False)
out["tracs_pairsnp_host_s"] = time.perf_counter() - t0
sel = (dh[np.triu_indices(n, 1)] <= 700)
assert len(r) == int(sel.sum()) and np.array_equal(dd.astype(np.int64), dh[r.astype(np.int64), c.astype(np.int64)])
assert np.array_equal(ncomp.astype(np.int64), nh[r.astype(np.int64), c.astype(np.int64)])
out["coo_pairs"] = int(len(r))
os.remove(path)
print(json.dumps(out))
