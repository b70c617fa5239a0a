"""partition.KeySplit over gloo on CPU (worlds 2, 3, 8): every rank holds only its own rows of the distance matrix (fold pairing, as
after the compact exchange), the distinct (N, day gap) keys of the WHOLE matrix are evaluated once -- each by one rank --, and every
rank's P / E(K) equal the single call's (the oracle's trans_dist over those cells).  tests/keysplit_standin.py stands in for the HIP
kernels (tests/test_gpu_keysplit.py holds the kernels against the stand-in); the protocol, the all-gathers and the decision to fall
back when the keys do not fit the grid are partition.KeySplit's own."""
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


def _matrix(n, seed, far):
    """a symmetric SNP matrix with structure (two clusters) and sampling days over two years; far: keys beyond the grid"""
    sys.path.insert(0, ROOT)
    from tracs_amd import synth
    rng = np.random.default_rng(seed)
    _, days = synth.dates(n, seed=seed)
    lab = rng.integers(0, 2, size=n)
    base = np.where(lab[:, None] == lab[None, :], 12, 90)
    d = base + rng.integers(0, 25, size=(n, n))
    if far:
        days = days.astype(np.int64) * 1500            # a span of ~10^6 days: (largest distance + 1) x (span + 1) beyond the grid's 2^24 keys
    d = np.triu(d, 1)
    return (d + d.T).astype(np.int32), days.astype(np.int32)


def _worker(rank, world, port, n, seed, far, thr, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import keysplit_standin as K
    from oracle import oracle as O
    from tracs_amd import partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lamb, beta, prec = 5.3, 6.0, 0.01
    calls = {"evaluated": 0}

    class Split(partition.KeySplit):
        def _words(self):
            return K.words()

        def _mark(self, dmat, days, ranges, keys, dist_threshold, col_begin):
            K.mark(dmat.numpy(), self.n, days.numpy(), ranges, keys.numpy().view(np.uint32), dist_threshold, col_begin)

        def _merge(self, keys, gathered):
            K.merge(keys.numpy().view(np.uint32), gathered.numpy().view(np.uint32), self.world)

        def _info(self, keys):
            return K.info(keys.numpy().view(np.uint32))

        def _evaluate(self, keys, info, lamb, beta, precision, vals):
            v = vals.numpy().reshape(-1, 2)
            K.evaluate(keys.numpy().view(np.uint32), info, self.rank, self.world, lamb, beta, precision, v, O.trans_dist)
            calls["evaluated"] += len(K.indices(keys.numpy().view(np.uint32))[self.rank::self.world])

        def _gather(self, dmat, days, ranges, keys, info, vals_all, pmat, emat, exp_p0, dist_threshold, col_begin):
            K.gather(dmat.numpy(), self.n, days.numpy(), ranges, keys.numpy().view(np.uint32), info, vals_all.numpy().reshape(self.world, -1, 2),
                     self.world, exp_p0, pmat.numpy(), emat.numpy(), dist_threshold, col_begin)

        def _whole(self, dmat, days, ranges, lamb, beta, precision, pmat, emat, exp_p0, dist_threshold, col_begin):
            d, dy = dmat.numpy(), days.numpy().astype(np.int64)
            for r0, r1 in ranges:
                for i in range(r0, r1):
                    j = np.arange(max(i + 1, col_begin), self.n)
                    j = j[d[i, j] <= dist_threshold]
                    p, e = O.trans_dist(d[i, j], (np.abs(dy[i] - dy[j]) * 86400).astype(np.float64) / 31556952.0, lamb, beta, precision)
                    pmat.numpy()[i, j] = np.exp(p) if exp_p0 else p
                    emat.numpy()[i, j] = e
    try:
        full, days = _matrix(n, seed, far)
        own = partition.own_row_ranges(0, n, rank, world, align=8)
        mine = np.zeros(n, dtype=bool)
        for r0, r1 in own:
            mine[r0:r1] = True
        dmat = torch.from_numpy(np.where(mine[:, None], full, -7).astype(np.int32))       # (rows of other ranks: garbage this rank must not read)
        pmat = torch.full((n, n), -1.0, dtype=torch.float64)
        emat = torch.full((n, n), -1.0, dtype=torch.float64)
        ks = Split(n, rank, world, dist, torch.device("cpu"))
        split = ks.run(dmat, torch.from_numpy(days), own, lamb, beta, prec, pmat, emat, exp_p0=True, dist_threshold=thr)
        i, j = np.triu_indices(n, 1)
        sel = mine[i] & (full[i, j] <= thr)
        rp, re = O.trans_dist(full[i[sel], j[sel]], (np.abs(days[i[sel]].astype(np.int64) - days[j[sel]]) * 86400).astype(np.float64) / 31556952.0,
                              lamb, beta, prec)
        ok = bool(np.array_equal(pmat.numpy()[i[sel], j[sel]], np.exp(rp)) and np.array_equal(emat.numpy()[i[sel], j[sel]], re))
        # nothing else was written: the rows of other ranks, the cells beyond the threshold, the lower triangle
        untouched = np.ones((n, n), dtype=bool)
        untouched[i[sel], j[sel]] = False
        ok = ok and bool((pmat.numpy()[untouched] == -1.0).all() and (emat.numpy()[untouched] == -1.0).all())
        whole = len({(int(a), int(b)) for a, b in zip(full[i, j][full[i, j] <= thr], np.abs(days[i].astype(np.int64) - days[j])[full[i, j] <= thr])})
        ret[rank] = (ok, split, ks.last_info, calls["evaluated"], whole, ks.bytes_gathered_per_call())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,far,thr", [(2, 61, False, 2147483647), (3, 40, False, 100), (8, 130, False, 2147483647), (8, 20, False, 2147483647),
                                             (2, 30, True, 2147483647)])
def test_key_split_equals_the_single_call(world, n, far, thr):
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, 20241022 + n, far, thr, ret), nprocs=world, join=True)
        got = dict(ret)
    assert len(got) == world and all(g[0] for g in got.values()), got
    routes = {g[1] for g in got.values()}
    assert routes == {not far}                                          # one decision, from the merged bitmap, on every rank
    assert len({g[2] for g in got.values()}) == 1                       # the same union everywhere
    if not far:
        keys = got[0][4]
        assert got[0][2][0] == keys                                      # the union holds the whole matrix's distinct keys
        assert sum(g[3] for g in got.values()) == keys                   # each evaluated by exactly one rank
        assert max(g[3] for g in got.values()) - min(g[3] for g in got.values()) <= 1
        per = -(-keys // world)
        assert got[0][5] == (world - 1) * ((2 ** 24 // 32 + 4) * 4 + per * 16)
