"""transcluster alone on a synthetic distance matrix: d ~ Poisson(mean) over all pairs of n samples, the bench's sampling days
(tracs_amd/synth.py) and `tracs distance` defaults.  usage: python scripts/probe_transcluster.py [mean_d ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import device as dev, synth  # noqa: E402

n = int(os.environ.get("N", "10000"))
_, days_np = synth.dates(n, seed=20241022)
days = torch.from_numpy(days_np).cuda()
g = torch.Generator(device="cuda")
g.manual_seed(7)
p = torch.empty((n, n), dtype=torch.float64, device="cuda")
e = torch.empty_like(p)
for mean in [float(x) for x in (sys.argv[1:] or ["1000", "10000", "20"])]:
    d = torch.poisson(torch.full((n, n), mean, device="cuda"), generator=g).to(torch.int32)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.trans_dist_dense_ranges(d, n, days, 1e-3 * 29903, 73.0, 0.01, p, e, [(0, n)], exp_p0=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    iu = torch.triu_indices(n, n, 1, device="cuda")
    print("mean d %6.0f: %7.2f ms  keys %d  checksum E(K) %.9e  P %.9e" % (mean, best * 1e3, dev._lib.load().tracs_debug_last_trans_dist_keys(),
                                                                       float(e[iu[0], iu[1]].sum()), float(p[iu[0], iu[1]].sum())))
