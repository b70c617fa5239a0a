"""The N > 1 path on CPU: 2, 3, 4 and 8 gloo ranks run the row-panel partition and the panel all-gather that bench.py / the
multi-GPU driver use (tracs_amd/partition.py) -- 32-bit panels and the 16-bit exchange --, with the oracle standing in for the kernel."""
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


def _worker(rank, world, port, n, L, seed, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tracs_amd import partition, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = synth.alignment(n, L, seed=seed, mu_lineage=0.02, mu_sample=0.01, p_n=0.03, p_partial=0.02)
        cs, nchunk = partition.row_chunks(n, world, align=8)
        dmat = torch.zeros((cs * nchunk, n), dtype=torch.int32)
        nmat = torch.zeros((cs * nchunk, n), dtype=torch.int32)
        planes = O.pack(seqs)
        mine = 0
        for r0, r1 in partition.rank_ranges(n, rank, world, align=8):
            # this rank's panel only: rows [r0, r1) against all later columns
            r, c, d, nn = O.pairsnp_planes(planes, L)
            sel = (r >= r0) & (r < r1)
            dmat[r[sel].astype(np.int64), c[sel].astype(np.int64)] = torch.from_numpy(d[sel].astype(np.int32))
            nmat[r[sel].astype(np.int64), c[sel].astype(np.int64)] = torch.from_numpy(nn[sel].astype(np.int32))
            mine += int(sel.sum())
            assert int(sel.sum()) == partition.pairs_in_rows(n, r0, r1)
        if n % 3 == 1:
            # the 16-bit exchange (partition.CompactPanels): d < 65 536 here; nn spans less than 65 536 around L
            cp = partition.CompactPanels(n, rank, world, dist, align=8)
            mode = cp.decide(dmat, nmat)
            assert mode[0] and mode[1] and cp.bytes_per_cell() == 4 and cp.check(dmat, nmat)
            cp.post(0, dmat, nmat, async_op=(n % 2 == 0))
            cp.finish(0, dmat, nmat)
        else:
            for w in partition.gather_panels((dmat, nmat), n, rank, world, dist, align=8, async_op=(n % 2 == 0)):
                w.wait()
        r, c, d, nn = O.pairsnp_planes(planes, L)
        full_d = torch.zeros((cs * nchunk, n), dtype=torch.int32)
        full_n = torch.zeros((cs * nchunk, n), dtype=torch.int32)
        full_d[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(d.astype(np.int32))
        full_n[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(nn.astype(np.int32))
        # (the 16-bit exchange carries the cells (i, j > i) only: whatever sits on or below the diagonal is not part of the result)
        up = torch.triu(torch.ones((cs * nchunk, n), dtype=torch.bool), diagonal=1)
        ok = bool(torch.equal(dmat[up], full_d[up]) and torch.equal(nmat[up], full_n[up]))
        ret[rank] = (ok, mine)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (2, 64), (3, 50), (4, 128), (4, 61), (8, 130), (8, 256)])
def test_partition_and_gather_gloo(world, n):
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, n, 300, 7, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert sum(ret[r][1] for r in range(world)) == n * (n - 1) // 2      # every pair owned by exactly one rank
    share = [ret[r][1] for r in range(world)]
    if n % (8 * 2 * world) == 0:                                          # chunks not distorted by the 8-row alignment
        assert max(share) - min(share) <= 0.02 * max(share)               # fold pairing balances the triangle


def test_partition_properties():
    from tracs_amd import partition
    for n in (1, 2, 63, 64, 65, 1000, 10000):
        for world in (1, 2, 4, 8):
            seen = np.zeros(n, int)
            tot = 0
            for rank in range(world):
                for r0, r1 in partition.rank_ranges(n, rank, world):
                    seen[r0:r1] += 1
                    tot += partition.pairs_in_rows(n, r0, r1)
            assert (seen == 1).all() and tot == n * (n - 1) // 2
    # 10k samples on 8 GPUs: per-rank work within 2 % of the mean
    w = [sum(partition.pairs_in_rows(10000, a, b) for a, b in partition.rank_ranges(10000, r, 8)) for r in range(8)]
    assert (max(w) - min(w)) / (sum(w) / 8) < 0.02


def _site_worker(rank, world, port, n, L, seed, ret):
    """Site shards (bench.py --partition sites, multigpu.pairs_site_sharded): rank r holds the sites of its whole 128-site groups,
    counts ALL pairs over them (the oracle standing in for the kernel), the partial matrices are summed over the ranks."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tracs_amd import multigpu, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = synth.alignment(n, L, seed=seed, mu_lineage=0.02, mu_sample=0.01, p_n=0.05, p_partial=0.02, p_other=0.01)
        groups = (L + 127) // 128
        g0, g1 = groups * rank // world, groups * (rank + 1) // world
        l0, l1 = g0 * 128, min(L, g1 * 128)
        cs = ((n + world - 1) // world + 7) // 8 * 8
        dmat = torch.zeros((cs * world, n), dtype=torch.int32)
        nmat = torch.zeros((cs * world, n), dtype=torch.int32)
        if l1 > l0:
            r, c, d, nn = O.pairsnp_arrays(seqs[:, l0:l1])
            dmat[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(d.astype(np.int32))
            nmat[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(nn.astype(np.int32))
        multigpu._sum_rows(dist, dmat, cs)
        multigpu._sum_rows(dist, nmat, cs)
        r, c, d, nn = O.pairsnp_arrays(seqs)
        ri, ci = r.astype(np.int64), c.astype(np.int64)
        own = (ri >= rank * cs) & (ri < (rank + 1) * cs)            # (what a reduce-scatter leaves on this rank; gloo sums everything)
        ok = bool(np.array_equal(dmat.numpy()[ri[own], ci[own]], d[own].astype(np.int32)) and
                  np.array_equal(nmat.numpy()[ri[own], ci[own]], nn[own].astype(np.int32)))
        ret[rank] = (ok, int(own.sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,L", [(2, 41, 700), (3, 50, 1000), (4, 33, 520), (8, 24, 300)])
def test_site_shards_sum_to_the_whole(world, n, L):
    """d and the compared-sites counts are sums over sites (src/pairsnp.hpp:398-403,417-420): ranks that each count a slice of the
    sites for all pairs, summed, give the whole alignment's matrices -- ragged last group, a rank without any site (8 ranks, 3
    groups), every IUPAC code."""
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_site_worker, args=(world, port, n, L, 20241022 + world, ret), nprocs=world, join=True)
        got = dict(ret)
    assert len(got) == world and all(ok for ok, _ in got.values()), got
    assert sum(k for _, k in got.values()) == n * (n - 1) // 2
