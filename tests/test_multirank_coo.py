"""The product-level N > 1 path on CPU (gloo): every rank computes the COO output of ITS row chunks, rank 0 assembles the
reference's row-major tuple (src/pairsnp.hpp:451-457) with partition.gather_coo; and the config-5 form -- per-rank threshold
edges on a transmission column -> gather -> connected components -- against SciPy on the whole graph.  The oracle stands in
for the kernels (no GPU here); the gather, the chunk ownership and the ordering are the code the GPU path runs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, L, thr, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tracs_amd import partition, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = synth.alignment(n, L, seed=11, mu_lineage=0.02, mu_sample=0.01, p_n=0.03, p_partial=0.02)
        _, days = synth.dates(n, seed=11, span_days=200)
        r, c, d, nn = O.pairsnp_arrays(seqs, dist=thr)                              # the whole thresholded output, row-major
        cs, _ = partition.row_chunks(n, world, align=8)
        # (1) `tracs distance`: this rank's chunks of (rows, cols, d, nn)
        parts, edges = {}, {}
        delta = np.abs(days[r.astype(np.int64)] - days[c.astype(np.int64)]).astype(np.float64) * 86400.0 / 31556952.0
        _, ek = O.trans_dist(d.astype(np.int32), delta, 5.3, 6.0, 0.01)
        keep = ek <= np.quantile(ek, 0.15)                                         # `tracs cluster -D expectedK -c T`: T keeps ~15 % of the pairs
        for ch in sorted(set(partition.rank_chunks(rank, world))):
            sel = (r >= ch * cs) & (r < min(n, (ch + 1) * cs))
            parts[ch] = tuple(torch.from_numpy(a[sel].astype(np.int32)) for a in (r, c, d, nn))
            edges[ch] = tuple(torch.from_numpy(a[sel & keep].astype(np.int32)) for a in (r, c))
        got = partition.gather_coo(parts, world, rank, dist)
        eg = partition.gather_coo(edges, world, rank, dist)
        if rank == 0:
            ok = all(np.array_equal(g.numpy().astype(np.uint64), a) for g, a in zip(got, (r, c, d, nn)))
            ok_e = np.array_equal(eg[0].numpy(), r[keep].astype(np.int32)) and np.array_equal(eg[1].numpy(), c[keep].astype(np.int32))
            from scipy.sparse import csr_matrix
            from scipy.sparse.csgraph import connected_components
            g = csr_matrix((np.ones(len(eg[0])), (eg[0].numpy(), eg[1].numpy())), shape=(n, n))
            ncomp, labels = connected_components(g, directed=False)
            g0 = csr_matrix((np.ones(int(keep.sum())), (r[keep].astype(np.int64), c[keep].astype(np.int64))), shape=(n, n))
            ncomp0, labels0 = connected_components(g0, directed=False)
            ret[0] = (ok, ok_e, ncomp == ncomp0 and np.array_equal(labels, labels0), int(keep.sum()), len(r))
        else:
            assert got is None and eg is None
            ret[rank] = (True,)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,thr", [(2, 41, 2147483647), (2, 64, 9), (3, 50, 12), (4, 70, 2147483647), (8, 129, 14)])
def test_coo_gather_and_edge_clustering_gloo(world, n, thr):
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, n, 300, thr, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok, ok_e, ok_cc, n_edges, n_pairs = ret[0]
    assert ok and ok_e and ok_cc
    assert n_pairs > 0 and 0 < n_edges < n_pairs


def test_gather_coo_single_rank_and_empty_chunks():
    import torch
    from tracs_amd import partition
    a = (torch.arange(5, dtype=torch.int32), torch.arange(5, dtype=torch.int32) + 10)
    e = (torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32))
    got = partition.gather_coo({0: a, 1: e}, 1, 0, None)
    assert got[0].tolist() == [0, 1, 2, 3, 4] and got[1].tolist() == [10, 11, 12, 13, 14]
    got = partition.gather_coo({0: e, 1: e}, 1, 0, None)
    assert got[0].numel() == 0 and got[0].dtype == torch.int32
    for world in (1, 2, 4, 8):
        owners = [partition.chunk_owner(c, world) for c in range(2 * world)]
        assert sorted(owners) == sorted(list(range(world)) * 2)
        for rank in range(world):
            assert all(partition.chunk_owner(c, world) == rank for c in partition.rank_chunks(rank, world))
