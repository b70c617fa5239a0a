#!/usr/bin/env python3
"""BASELINE.json configs[4] at shape: N samples, transcluster K-integral on every pair + single-linkage threshold clustering.

    python scripts/bench_config5.py [--samples 100000] [--gpus N]      (N > 1: python -m torch.distributed.run ... like bench.py)

SNP distances are synthetic (SURVEY.md 8d's two-component mixture: Poisson(3) for close pairs, else Poisson(80), capped by -D 100;
dates over 730 days; lambda = 5.3, beta = 6, precision 0.01; edges where E(K) <= 5).  WHICH pairs are close is structured like
outbreaks, not drawn per pair (iid close pairs at SURVEY's rate 0.001 give every sample ~100 random neighbours and one giant
component, which no labelling bug could hide behind): every sample belongs to one of --clusters transmission clusters (a hash of
its index), pairs inside a cluster are close, and one pair in 10^6 elsewhere is close too (occasional links between clusters).
With the defaults the E(K) <= 5 graph has tens of thousands of components -- clusters split further by sampling date -- and the
labels of ALL samples are checked against SciPy.  Every rank walks ITS row chunks in bounded panels:
generate d for the panel -> tracs_trans_dist_dense -> tracs_edges_*_f64 -> only the (i, j) of surviving edges leave the device;
rank 0 gathers the edge lists (partition.gather_coo: counts, then variable-length payloads) and labels the components.
The dense N x N matrices never exist (at 100 000 samples they would be 40 GB each).  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--clusters", type=int, default=10000, help="transmission clusters the samples are hashed into")
    ap.add_argument("--check", type=int, default=-1, help="SciPy check of the labels on the first CHECK samples (rank 0); -1: all of them")
    args = ap.parse_args()
    import numpy as np
    import torch
    from tracs_amd import multigpu, partition, synth
    from tracs_amd import device as dev
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist, rank, world, device = multigpu.init()
    else:
        dist, rank, device = None, 0, torch.device("cuda", 0)
        torch.cuda.set_device(0)
    n = args.samples
    _, days_np = synth.dates(n, seed=20241022 + 4)
    days = torch.from_numpy(days_np).to(device)

    def synth_panel(dpan, npan, r0, r1):
        """d(i, j) from a counter-based hash of (i, j): the same value whichever rank or panel computes it."""
        rows = torch.arange(r0, r1, device=device, dtype=torch.int64)[:, None]
        cols = torch.arange(0, n, device=device, dtype=torch.int64)[None, :]
        h = (rows * 0x1E3779B97F4A7C15 + cols * 0x42B2AE3D27D4EB4F) & 0x7FFFFFFFFFFFFFFF
        h = (h ^ (h >> 29)) * 0x3F58476D1CE4E5B9 & 0x7FFFFFFFFFFFFFFF
        h = h ^ (h >> 32)
        u = (h & 0xFFFFFF).double() / float(1 << 24)
        def cluster_of(v):
            c = (v * 0x2545F4914F6CDD1D) & 0x7FFFFFFFFFFFFFFF
            return ((c ^ (c >> 31)) * 0x9E3779B1 & 0x7FFFFFFFFFFFFFFF) % args.clusters
        close = (cluster_of(rows) == cluster_of(cols)) | (((h >> 24) & 0xFFFFF) == 0)   # same cluster, or one pair in 2^20
        mean = torch.where(close, torch.tensor(3.0, device=device, dtype=torch.float64), torch.tensor(80.0, device=device, dtype=torch.float64))
        # Poisson by the normal approximation around the mean is enough for a workload shape; deterministic in (i, j)
        z = torch.erfinv(2.0 * u.clamp(1e-7, 1 - 1e-7) - 1.0) * 1.4142135623730951
        d = torch.clamp(torch.round(mean + z * torch.sqrt(mean)), 0, 100).to(torch.int32)
        dpan[:r1 - r0] = d
        npan[:r1 - r0] = 1

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    parts = multigpu.edges_of_rank(synth_panel, n, days, 5.3, 6.0, 0.01, "expectedK", 5.0, rank, world, dist_threshold=100)
    torch.cuda.synchronize()
    t_edges = time.perf_counter() - t0
    res = multigpu.cluster_edges(parts, n, rank, world, dist) if world > 1 else None
    if world == 1:
        got = partition.gather_coo(parts, 1, 0, None)
        res = dev.connected_components_device(got[0], got[1], n)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_all = time.perf_counter() - t0
    if rank == 0:
        ncomp, labels = res
        n_edges = sum(int(p[0].numel()) for p in parts.values()) if world == 1 else None
        out = {"workload": "config 5 shape: %d samples, synthetic SNP distances, transcluster on every pair, E(K) <= 5 edges, connected components" % n,
               "n_gpus": world, "pairs": n * (n - 1) // 2, "seconds": t_all, "seconds_edges": t_edges,
               "pairs_per_s": n * (n - 1) / 2 / t_all, "components": int(ncomp), "edges_rank0_chunks": n_edges, "clusters": args.clusters}
        if world == 1:
            lab = labels.cpu().numpy()
            sizes = np.bincount(lab)
            out["largest_component"] = int(sizes.max())
            out["components_of_size"] = {"1": int((sizes == 1).sum()), "2-5": int(((sizes >= 2) & (sizes <= 5)).sum()),
                                         "6-20": int(((sizes >= 6) & (sizes <= 20)).sum()), ">20": int((sizes > 20).sum())}
        if args.check and world == 1:
            from scipy.sparse import csr_matrix
            from scipy.sparse.csgraph import connected_components
            m = n if args.check < 0 else min(args.check, n)
            i, j = got[0].cpu().numpy(), got[1].cpu().numpy()
            sel = (i < m) & (j < m)
            g = csr_matrix((np.ones(int(sel.sum()), np.int8), (i[sel], j[sel])), shape=(m, m))
            # components of the induced subgraph on the first m samples, recomputed on the GPU for the same edge subset
            nc2, lab2 = dev.connected_components_device(torch.from_numpy(i[sel]).to(device), torch.from_numpy(j[sel]).to(device), m)
            enc, elab = connected_components(g, directed=False)
            out["scipy_check"] = bool(enc == nc2 and np.array_equal(elab, lab2.cpu().numpy()))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
