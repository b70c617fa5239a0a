"""The HIP kernels behind partition.KeySplit (csrc/transcluster.hip tracs_trans_keys_*) on one GPU: word for word against
tests/keysplit_standin.py, and -- P ranks played one after the other in this process -- P / E(K) of every rank's own rows BIT-EQUAL
to the single call over the whole matrix (tracs_trans_dist_dense2), each distinct key evaluated by exactly one rank."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _matrix(n, seed, spread=25, scale=1):
    from tracs_amd import synth
    rng = np.random.default_rng(seed)
    _, days = synth.dates(n, seed=seed)
    lab = rng.integers(0, 3, size=n)
    d = np.where(lab[:, None] == lab[None, :], 12, 90) + rng.integers(0, spread, size=(n, n))
    d = np.triu(d * scale, 1)
    return (d + d.T).astype(np.int32), days.astype(np.int32)


@pytest.mark.parametrize("n,ranges,thr,col_begin", [(300, [(0, 300)], 2147483647, 0), (333, [(8, 40), (290, 333)], 100, 0), (200, [(0, 64)], 2147483647, 64),
                                                    (130, [], 2147483647, 0)])
def test_mark_merge_info_match_the_standin(n, ranges, thr, col_begin):
    import torch
    import keysplit_standin as K
    from tracs_amd import device as dev
    assert dev.trans_keys_words() == K.words()
    d, days = _matrix(n, 5 + n)
    dm, dy = torch.from_numpy(d).cuda(), torch.from_numpy(days).cuda()
    keys = torch.full((K.words(),), -1, dtype=torch.int32, device="cuda")
    dev.trans_keys_mark(dm, n, dy, ranges, keys, thr, col_begin)
    want = np.zeros(K.words(), dtype=np.uint32)
    K.mark(d, n, days, ranges, want, thr, col_begin)
    got = keys.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert dev.trans_keys_info(keys) == K.info(want)
    # merge: three bitmaps (other row ranges, another threshold) end to end
    parts = [want]
    allk = [keys]
    for r, t in (([(0, n // 2)], 2147483647), ([(n // 3, n)], 95)):
        k2 = torch.empty_like(keys)
        dev.trans_keys_mark(dm, n, dy, r, k2, t, 0)
        w2 = np.zeros(K.words(), dtype=np.uint32)
        K.mark(d, n, days, r, w2, t, 0)
        assert np.array_equal(k2.cpu().numpy().view(np.uint32), w2)
        parts.append(w2); allk.append(k2)
    u = torch.empty_like(keys)
    dev.trans_keys_merge(u, torch.cat(allk).contiguous(), 3)
    wu = np.zeros(K.words(), dtype=np.uint32)
    K.merge(wu, np.concatenate(parts), 3)
    assert np.array_equal(u.cpu().numpy().view(np.uint32), wu)
    assert dev.trans_keys_info(u) == K.info(wu)


@pytest.mark.parametrize("n,world,thr", [(257, 1, 2147483647), (500, 2, 2147483647), (401, 3, 100), (1000, 8, 2147483647), (40, 8, 2147483647), (3000, 8, 2147483647)])
def test_split_keys_equal_the_single_call(n, world, thr):
    import torch
    import keysplit_standin as K
    from tracs_amd import _lib
    from tracs_amd import device as dev
    from tracs_amd import partition
    lamb, beta, prec = 5.3, 6.0, 0.01
    d, days = _matrix(n, 11 + n, spread=40 if n < 2000 else 400)
    dm, dy = torch.from_numpy(d).cuda(), torch.from_numpy(days).cuda()
    p1 = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    e1 = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    dev.trans_dist_dense_ranges(dm, n, dy, lamb, beta, prec, p1, e1, [(0, n)], exp_p0=True, dist_threshold=thr)
    keys_whole = int(_lib.load().tracs_debug_last_trans_dist_keys())
    own = [partition.own_row_ranges(0, n, q, world) for q in range(world)]
    words = dev.trans_keys_words()
    gathered = torch.empty(world * words, dtype=torch.int32, device="cuda")
    for q in range(world):
        # (a rank sees only its rows: the others' are garbage to it)
        mine = torch.zeros(n, dtype=torch.bool, device="cuda")
        for r0, r1 in own[q]:
            mine[r0:r1] = True
        dq = torch.where(mine[:, None], dm, torch.full_like(dm, 77777))
        dev.trans_keys_mark(dq, n, dy, own[q], gathered[q * words:(q + 1) * words], thr)
    union = torch.empty(words, dtype=torch.int32, device="cuda")
    dev.trans_keys_merge(union, gathered, world)
    info = dev.trans_keys_info(union)
    assert info[3] == 1 and info[0] == keys_whole                     # the union = the whole matrix's distinct keys
    per = max(1, -(-info[0] // world))
    vals_all = torch.full((world * per * 2,), float("nan"), dtype=torch.float64, device="cuda")
    evaluated = 0
    for q in range(world):
        dev.trans_keys_evaluate(union, info, q, world, lamb, beta, prec, vals_all[q * per * 2:(q + 1) * per * 2])
        evaluated += int(_lib.load().tracs_debug_last_trans_dist_keys())
    assert evaluated == info[0]                                        # each key by exactly one rank
    # the compact arrays against the oracle-free stand-in's numbering: slot (o % P, o // P) holds the key of ordinal o
    idx = K.indices(union.cpu().numpy().view(np.uint32))
    assert len(idx) == info[0]
    va = vals_all.cpu().numpy().reshape(world, per, 2)
    o = np.arange(len(idx))
    assert not np.isnan(va[o % world, o // world]).any()
    for q in range(world):
        p0 = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
        ek = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
        mine = torch.zeros(n, dtype=torch.bool, device="cuda")
        for r0, r1 in own[q]:
            mine[r0:r1] = True
        dq = torch.where(mine[:, None], dm, torch.full_like(dm, 77777))
        dev.trans_keys_gather(dq, n, dy, own[q], union, info, vals_all, world, p0, ek, exp_p0=True, dist_threshold=thr)
        assert torch.equal(p0[mine], p1[mine]) and torch.equal(ek[mine], e1[mine])       # bit-equal, incl. the untouched cells (-1)
        assert bool((p0[~mine] == -1.0).all())


def test_keys_beyond_the_grid_fall_back():
    import torch
    from tracs_amd import device as dev
    from tracs_amd import partition
    n = 200
    d, days = _matrix(n, 3, scale=40000)                               # distances of millions: N x span beyond 2^24 keys
    dm, dy = torch.from_numpy(d).cuda(), torch.from_numpy(days).cuda()
    keys = torch.empty(dev.trans_keys_words(), dtype=torch.int32, device="cuda")
    dev.trans_keys_mark(dm, n, dy, [(0, n)], keys)
    info = dev.trans_keys_info(keys)
    assert info[3] == 0
    with pytest.raises(RuntimeError, match="do not fit the grid"):
        dev.trans_keys_evaluate(keys, info, 0, 1, 5.3, 6.0, 0.01, torch.zeros(2 * max(1, info[0]), dtype=torch.float64, device="cuda"))
    # KeySplit on a world of one rank: falls back to the whole evaluation and says so
    d2, days2 = _matrix(n, 4)
    dy2 = torch.from_numpy(days2.astype(np.int64) * 1500).to(torch.int32).cuda()      # a span of ~10^6 days
    dm2 = torch.from_numpy(d2).cuda()
    pa, ea = torch.zeros((n, n), dtype=torch.float64, device="cuda"), torch.zeros((n, n), dtype=torch.float64, device="cuda")
    pb, eb = torch.zeros_like(pa), torch.zeros_like(ea)
    ks = partition.KeySplit(n, 0, 1, None, torch.device("cuda"))
    assert ks.run(dm2, dy2, [(0, n)], 5.3, 6.0, 0.01, pa, ea) is False and ks.last_route == "whole"
    dev.trans_dist_dense_ranges(dm2, n, dy2, 5.3, 6.0, 0.01, pb, eb, [(0, n)], exp_p0=True)
    assert torch.equal(pa, pb) and torch.equal(ea, eb)
    # ... and splits when the keys fit
    dy3 = torch.from_numpy(days2).cuda()
    assert ks.run(dm2, dy3, [(0, n)], 5.3, 6.0, 0.01, pa, ea) is True and ks.last_route == "split"
    dev.trans_dist_dense_ranges(dm2, n, dy3, 5.3, 6.0, 0.01, pb, eb, [(0, n)], exp_p0=True)
    assert torch.equal(pa, pb) and torch.equal(ea, eb)
