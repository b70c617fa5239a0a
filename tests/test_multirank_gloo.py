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


def _site_worker(rank, world, port, n, L, seed, n0, ret):
    """Site shards (bench.py --partition sites, multigpu.pairs_site_sharded): rank r holds the sites of its whole 128-site groups,
    counts ALL pairs over them (the oracle standing in for the kernel), and the partial matrices are summed by the compact exchange
    -- partition.TriExchange's own layout, width decision, all-to-all and protocol, with tests/tri_standin.py standing in for the two
    HIP kernels of csrc/exchange.hip: upper-triangle cells only, 16 bits per cell where the slice's values fit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import tri_standin
    from oracle import oracle as O
    from tracs_amd import partition, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Tri(partition.TriExchange):
        def _pack(self, mat, base_row, slots, width, base, negate, packed_ptr, stats):
            st = stats.numpy().view(np.uint32)
            tri_standin.tri_pack(mat.numpy().view(np.uint32), self.n, self.rb, self.re, self.cb, slots.numpy(), width, base, negate, packed_ptr,
                                 self.world * self.block_elems * 4, st, base_row)

        def _sum(self, mat, base_row, slots, width, recv_ptr, block_elems, add, negate):
            tri_standin.tri_sum(mat.numpy().view(np.uint32), self.n, self.rb, self.re, self.cb, slots.numpy(), width, recv_ptr,
                                self.world * block_elems, block_elems, self.world, self.rank, add, negate, base_row)
    try:
        # wide == True: distances and deficits beyond 16 bits (every sample far from every other, a third of the sites N)
        wide = L > 100000
        seqs = synth.alignment(n, L, seed=seed, mu_lineage=0.02, mu_sample=0.6 if wide else 0.01, p_n=0.35 if wide else 0.05,
                               p_partial=0.02, p_other=0.01)
        groups = (L + 127) // 128
        g0, g1 = groups * rank // world, groups * (rank + 1) // world
        l0, l1 = g0 * 128, min(L, g1 * 128)
        i_end, j_start = (n, 0) if n0 is None else (n0, n0)           # two-file mode: rows of the first file x columns of the second
        dmat = torch.zeros((n, n), dtype=torch.int32)
        nmat = torch.zeros((n, n), dtype=torch.int32)
        if l1 > l0:
            r, c, d, nn = O.pairsnp_arrays(seqs[:, l0:l1], n0=n0)
            dmat[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(d.astype(np.int32))
            nmat[r.astype(np.int64), c.astype(np.int64)] = torch.from_numpy(nn.astype(np.int32))
        ex = Tri(n, 0, i_end, j_start, rank, world, dist, torch.device("cpu"), align=8)
        widths = ex.decide(dmat, nmat, l1 - l0)
        ex.run(dmat, nmat, l1 - l0, L)
        r, c, d, nn = O.pairsnp_arrays(seqs, n0=n0)
        ri, ci = r.astype(np.int64), c.astype(np.int64)
        own = np.zeros(len(ri), dtype=bool)
        for q0, q1 in ex.own_ranges:
            own |= (ri >= q0) & (ri < q1)
        ok = bool(np.array_equal(dmat.numpy()[ri[own], ci[own]], d[own].astype(np.int32)) and
                  np.array_equal(nmat.numpy()[ri[own], ci[own]], nn[own].astype(np.int32)) and ex.check())
        ret[rank] = (ok, int(own.sum()), widths, ex.bytes_sent_per_call(), ex.own_cells)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,L,n0", [(2, 41, 700, None), (3, 50, 1000, None), (4, 33, 520, 9), (8, 24, 300, None), (2, 12, 400000, None), (3, 10, 900000, None)])
def test_site_shards_sum_to_the_whole(world, n, L, n0):
    """d and the compared-sites counts are sums over sites (src/pairsnp.hpp:398-403,417-420): ranks that each count a slice of the
    sites for all pairs, summed by the compact exchange, give the whole alignment's matrices on the rows each rank owns -- ragged
    last group, a rank without any site (8 ranks, 3 groups), every IUPAC code, two-file mode (rows of the first file only), and both
    widths: 16 bits per cell where the slice's values fit, 32 where they do not (the last two cases: nn only, then both)."""
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_site_worker, args=(world, port, n, L, 20241022 + world, n0, ret), nprocs=world, join=True)
        got = dict(ret)
    assert len(got) == world and all(g[0] for g in got.values()), got
    pairs = n * (n - 1) // 2 if n0 is None else n0 * (n - n0)
    assert sum(g[1] for g in got.values()) == pairs and sum(g[4] for g in got.values()) == pairs
    widths = {g[2] for g in got.values()}
    assert len(widths) == 1                                             # agreed over the ranks
    assert widths == {{400000: (2, 4), 900000: (4, 4)}.get(L, (2, 2))}, widths     # (mixed widths: the blocks' stride is not a multiple of the wider cell)
    # what a rank sends: P - 1 blocks of the largest share of the cells, 4 bytes per cell -- a quarter of the two full uint32
    # matrices round 4 reduce-scattered when the shares are equal (they are not at these sizes: chunks of 8 rows)
    if L <= 100000 and n0 is None and world <= 3:
        assert all(g[3] <= 0.4 * 2 * 4 * n * n * (world - 1) / world for g in got.values()), got


def test_tri_layout_at_bench_size():
    """The compact exchange's layout at 10 000 samples: every cell (i, j > i) has exactly one owner and one slot, the ranks' shares
    are equal to within a chunk of rows, and what a rank sends per call -- P - 1 blocks, 2 + 2 bytes per cell -- stays under 0.2 GB
    (round 4 reduce-scattered two full uint32 matrices: 0.7 GB per rank at P = 8)."""
    sys.path.insert(0, ROOT)
    from tracs_amd import partition
    n = 10000
    for world in (2, 4, 8):
        cs, owner, off, be = partition.tri_layout(0, n, n, 0, world)
        cells = n - 1 - np.arange(n)
        assert cs % 64 == 0 and be % 64 == 0
        share = np.array([cells[owner == q].sum() for q in range(world)])
        assert share.sum() == n * (n - 1) // 2 and share.max() <= be < share.max() + 64
        assert share.max() - share.min() <= 2 * cs * n // world + cs * cs          # fold pairing: equal to within the clipped last chunk
        for q in range(world):
            sel = np.nonzero(owner == q)[0]
            assert np.array_equal(off[sel], np.cumsum(cells[sel]) - cells[sel])     # rows of a block back to back, ascending
            rng = partition.own_row_ranges(0, n, q, world)
            assert sorted(sel.tolist()) == [i for a, b in rng for i in range(a, b)]
        sent = (world - 1) * be * 4
        assert sent <= 0.2e9, (world, sent)
        assert sent <= 0.27 * 2 * 4 * n * n * (world - 1) / world                   # a quarter of round 4's bytes (+ the ragged last chunk)
