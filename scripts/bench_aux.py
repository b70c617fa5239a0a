"""Roofline measurements of the secondary kernels: posteriors (f64 and fused u16 -> code),
transcluster per-pair path at config-5 shape, connected components, pack, recombination filter.
Prints one JSON object; numbers go to DESIGN.md / profiles/.  usage: python scripts/bench_aux.py [scale]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
d = torch.device("cuda", 0)
out = {}


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for r in range(reps):
        ev[r].record()
        fn()
    ev[reps].record()
    torch.cuda.synchronize()
    return min(ev[r].elapsed_time(ev[r + 1]) for r in range(reps)) / 1e3


alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
# --- posteriors: config 4 is 50 000 samples x 1 Mbp; one launch here = 200 Mbp of sites (200 samples' worth)
L = int(200e6 * scale)
cnt16 = torch.randint(0, 40, (L, 4), device=d, dtype=torch.int16)
t = timed(lambda: dev.posterior_codes_device(cnt16, alphas, False, 0.01))
out["posterior_codes_u16"] = {"sites": L, "s": t, "alg_bytes_per_site": 8.5, "GBps": L * 8.5 / t / 1e9, "frac_of_8TBps": L * 8.5 / t / 8e12}
Lf = int(50e6 * scale)
cntf = torch.randint(0, 40, (Lf, 4), device=d).to(torch.float64)
t = timed(lambda: dev.calculate_posteriors_device(cntf, alphas, False, 0.01))
out["calculate_posteriors_f64"] = {"sites": Lf, "s": t, "alg_bytes_per_site": 64, "GBps": Lf * 64 / t / 1e9, "frac_of_8TBps": Lf * 64 / t / 8e12}
del cnt16, cntf
# --- transcluster, config-5 shape: SNP distances from a two-component mixture, day gaps, lambda=5.3 beta=6
P = int(200e6 * scale)
g = torch.Generator(device=d); g.manual_seed(5)
close = torch.rand(P, generator=g, device=d) < 0.001
N = torch.where(close, torch.poisson(torch.full((P,), 3.0, device=d), generator=g), torch.clamp(torch.poisson(torch.full((P,), 80.0, device=d), generator=g), max=100)).to(torch.int32)
days = torch.randint(0, 730, (P,), generator=g, device=d)
delta = days.to(torch.float64) * 86400.0 / 31556952.0
t0 = time.perf_counter(); p0, ek = dev.trans_dist_device(N, delta, 5.3, 6.0, 0.01, exp_p0=True); torch.cuda.synchronize(); first = time.perf_counter() - t0
t = timed(lambda: dev.trans_dist_device(N, delta, 5.3, 6.0, 0.01, exp_p0=True), reps=3)
nkeys = int(torch.unique(N.to(torch.int64) * 1000 + days).numel())
out["trans_dist_device"] = {"pairs": P, "distinct_keys": nkeys, "s": t, "first_call_s": first, "pairs_per_s": P / t,
                            "alg_bytes_per_pair": 28, "GBps": P * 28 / t / 1e9, "frac_of_8TBps": P * 28 / t / 8e12}
# threshold + clustering on 100 000 samples' worth of edges
thr_edges = (ek <= 5.0).nonzero().flatten()
n_nodes = 100000
gi = torch.randint(0, n_nodes, (thr_edges.numel(),), generator=g, device=d, dtype=torch.int32)
gj = torch.randint(0, n_nodes, (thr_edges.numel(),), generator=g, device=d, dtype=torch.int32)
t = timed(lambda: dev.connected_components_device(gi, gj, n_nodes))
nc, lab = dev.connected_components_device(gi, gj, n_nodes)
out["connected_components"] = {"nodes": n_nodes, "edges": int(gi.numel()), "components": nc, "s": t,
                               "GBps_8B_per_edge": gi.numel() * 8 / t / 1e9}
del N, days, delta, p0, ek, gi, gj
# --- pack: 64 samples x 5 Mbp of ASCII already on the device
Lp, ns = 5000000, 64
asc = torch.randint(65, 85, (ns, Lp), device=d, dtype=torch.uint8)
aln = dev.Alignment(ns, Lp)
t = timed(lambda: aln.pack(asc, first=0))
out["pack_kernel"] = {"bytes_in": ns * Lp, "s": t, "alg_bytes_per_site": 1.625, "GBps": ns * Lp * 1.625 / t / 1e9,
                      "frac_of_8TBps": ns * Lp * 1.625 / t / 8e12}
print(json.dumps(out))
