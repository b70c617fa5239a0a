"""The lists of an alignment cut into site classes (csrc/site_lists.hip), structure by structure against numpy: the n8 lines of
the per-site N lists (byte deltas, sorted, 124 payload bytes + next line), the p lists, the per-sample listed entries, the
rows' N bitmaps.  The pair results built on them are checked against the oracle in test_gpu_site_classes.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODES = {c: m for c, m in zip(b"ACGTMRWSYKVHDB", (1, 2, 4, 8, 3, 5, 9, 6, 10, 12, 7, 11, 13, 14))}


def _masks(seqs):
    up = np.where((seqs >= 97) & (seqs <= 122), seqs - 32, seqs)
    lut = np.full(256, 15, dtype=np.uint8)
    for c, m in CODES.items():
        lut[c] = m
    return lut[up]


def _dump(lib, aln, what, dtype, count):
    buf = np.empty(count, dtype=dtype)
    got = lib.tracs_debug_lists(aln._h, what, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert got == buf.nbytes, (what, got, buf.nbytes)
    return buf


def _decode(lines, r):
    """the samples of the list whose primary line is r (reference decoder of the n8 format, csrc/pairsnp_kernels.h)"""
    out, p, line, hops = [], -1, int(r), 0
    while True:
        payload = lines[line, :124]
        pad = np.nonzero(payload == 255)[0]
        if pad.size:
            assert (payload[pad[0]:] == 255).all(), "padding inside a line"
            assert (lines[line, 124:] == 255).all(), "a padded line goes on"
        for b in payload.tolist():
            if b < 253:
                p += b
                out.append(p)
            elif b == 253:
                p += 253
            else:
                assert b == 255
        nxt = int(lines[line, 124:].view(np.uint32)[0])
        if nxt == 0xFFFFFFFF:
            return out
        line, hops = nxt, hops + 1
        assert hops < 100000


CASES = [
    dict(n=640, L=5000, p_n=0.02, mu=3e-4, seed=1, bitmaps=True),        # short lists, one piece
    dict(n=700, L=3000, p_n=0.30, mu=3e-4, seed=2, bitmaps=True),       # ~210 N samples per site: two lines per list, several pieces per group
    dict(n=2000, L=1500, p_n=0.002, mu=2e-4, seed=3, bitmaps=True),     # gaps beyond 253: skip bytes
    dict(n=130, L=4000, p_n=0.05, mu=2e-3, seed=4, p_partial=0.002),    # partial codes among the listed samples
    dict(n=700, L=3000, p_n=0.01, mu=3e-4, seed=5, bitmaps=True, edges=True),      # lists that end exactly at the encoder's boundaries
    dict(n=900, L=6000, p_n=0.01, mu=6e-3, seed=6, bitmaps=True),       # ~30 000 listed entries: the bucketed fill of the per-sample lists
    dict(n=4000, L=1024, p_n=0.001, mu=1e-4, seed=7, p_partial=0.009, long_p=True, qw=64),      # ~36 partial codes per site: 256-byte q lines, some lists beyond one
    dict(n=4000, L=1024, p_n=0.001, mu=1e-4, seed=7, p_partial=0.009, long_p=True, qw=32, env={"TRACS_QLINE_DWORDS": "32"}),      # ... the same in 128-byte q lines: most lists with an overflow line
    dict(n=4000, L=1024, p_n=0.0005, mu=1e-4, seed=8, p_partial=0.019, long_p=True),     # ~76 per site, ~9 700 per group: beyond the LDS image of p_lists_kernel (8 192)
    dict(n=2000, L=2048, p_n=0.002, mu=1e-4, seed=9, p_partial=0.0055),                  # ~11 per site, ~1 400 per group: p_lists_kernel with short (<= 4) and long lists side by side
]

# N samples of hand-made sites (case `edges`): the byte counts the encoder's branches turn on
EDGE_LISTS = [
    list(range(124)),            # 124 bytes: a full line and nothing behind it (no padding, no next line)
    list(range(123)),            # one byte short of it
    list(range(125)),            # one byte into a follow-on line
    list(range(248)),            # two full lines exactly
    list(range(16)),             # one piece exactly
    list(range(112)),            # seven pieces exactly: the line's last piece is padding + "no next line"
    [0, 300, 600],               # gaps beyond 253: skip bytes
    [252, 505, 699],             # gaps of exactly 253: a skip byte, then a byte 0
    [5, 699],                    # the last sample
    list(range(0, 700, 2)),      # 350 bytes of 2: three lines
]


@pytest.mark.parametrize("nn_lists", ["always", "cost-model"])
@pytest.mark.parametrize("case", range(len(CASES)), ids=lambda k: "c%d_n%d_L%d_pn%g" % (k, CASES[k]["n"], CASES[k]["L"], CASES[k]["p_n"]))
def test_lists_against_numpy(hiplib, case, nn_lists):
    """(a child process per case: the list threshold TRACS_NN_LIST_K is read once per process)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("TRACS_NN_LIST_K", None)
    env.pop("TRACS_QLINE_DWORDS", None)
    if nn_lists == "always":
        env["TRACS_NN_LIST_K"] = "1"
    env.update(CASES[case].get("env", {}))
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_lists as T; T.run_case(T.CASES[%d], %r)" % (
        root, os.path.join(root, "tests"), case, nn_lists)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]


def run_case(case, nn_lists):
    import torch
    from tracs_amd import _lib, device as dev
    hiplib = _lib.load()
    n, L = case["n"], case["L"]
    rng = np.random.default_rng(case["seed"])
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = np.tile(bases[rng.integers(0, 4, size=L)], (n, 1))
    mut = rng.random((n, L)) < case["mu"]
    seqs[mut] = bases[rng.integers(0, 4, size=int(mut.sum()))]
    seqs[rng.random((n, L)) < case["p_n"]] = ord("N")
    if case.get("p_partial"):
        part = rng.random((n, L)) < case["p_partial"]
        seqs[part] = np.frombuffer(b"MRWSYKVHDB", dtype=np.uint8)[rng.integers(0, 10, size=int(part.sum()))]
    seqs[:, rng.random(L) < 0.01] = ord("N")                              # empty sites
    if case.get("edges"):
        for k, samples in enumerate(EDGE_LISTS):
            t = 200 + 37 * k                                              # (sites of several groups)
            seqs[:, t] = ord("A")
            seqs[samples, t] = ord("N")
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros_like(d)
    try:
        hiplib.tracs_debug_force_site_classes(1)
        dev.pairsnp_dense(aln, d, nn)
    finally:
        hiplib.tracs_debug_force_site_classes(-2)
    assert aln.site_classes is not None
    from oracle import oracle as O
    er, ec, ed, enn = O.pairsnp_arrays(seqs, n_threads=8)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    probe = np.empty(8, dtype=np.uint64)
    if hiplib.tracs_debug_lists(aln._h, 0, probe.ctypes.data_as(C.c_void_p), 64) != 64:
        # the lists were refused (they would outweigh the planes): classes without lists -- nothing to look at
        assert not case.get("bitmaps") or nn_lists != "always"
        print("lists refused", flush=True)
        aln.close()
        return
    sizes = _dump(hiplib, aln, 0, np.uint64, 8)
    sites, n_lines, tot_p, tgroups, groups, has_T = (int(x) for x in sizes[:6])
    assert sites > 0 and groups == (L + 127) // 128
    lines = _dump(hiplib, aln, 1, np.uint8, n_lines * 128).reshape(n_lines, 128)
    lst_mask = _dump(hiplib, aln, 2, np.uint32, groups * 4)
    off_lst = _dump(hiplib, aln, 3, np.uint32, groups)
    listed_site = np.unpackbits(lst_mask.view(np.uint8), bitorder="little")[:L].astype(bool)
    assert int(listed_site.sum()) == sites
    rank = np.cumsum(listed_site) - 1
    before = np.concatenate([[0], np.cumsum(listed_site)])
    assert (off_lst == before[np.minimum(np.arange(groups) * 128, L)]).all()
    M = _masks(seqs)
    isN = M == 15
    if case.get("edges") and nn_lists == "always":
        assert all(listed_site[200 + 37 * k] for k in range(len(EDGE_LISTS))), "a hand-made site has no list"
    # ---- N lists: every listed site's line chain decodes to its N samples, in order
    for t in np.nonzero(listed_site)[0]:
        got = _decode(lines, rank[t])
        want = np.nonzero(isN[:, t])[0].tolist()
        assert got == want, (t, got[:10], want[:10])
    # ---- p lists and the per-sample listed entries
    p_off = _dump(hiplib, aln, 4, np.uint64, sites + 1)
    assert int(p_off[-1]) == tot_p
    if tot_p:
        p_ent = _dump(hiplib, aln, 5, np.uint32, tot_p)                              # lists of at most 4 samples
        nq = np.zeros(1, np.uint64)
        assert hiplib.tracs_debug_lists(aln._h, 11, nq.ctypes.data_as(C.c_void_p), 8) == 8
        ks_all = np.diff(p_off.astype(np.int64))
        assert (int(nq[0]) >= sites) == bool((ks_all > 4).any())                     # q lines exist iff some list is a long one
        qwb = np.zeros(1, np.uint64)
        assert hiplib.tracs_debug_lists(aln._h, 12, qwb.ctypes.data_as(C.c_void_p), 8) == 8
        qw = int(qwb[0])                                                             # dwords per q line: 32, or 64 when the lists are long on average
        if case.get("env", {}).get("TRACS_QLINE_DWORDS"):
            assert qw == int(case["env"]["TRACS_QLINE_DWORDS"])
        else:
            assert qw == (64 if ks_all[ks_all > 0].mean() > 24 and (ks_all > 4).any() else 32), (qw, ks_all[ks_all > 0].mean())
        if case.get("qw"):
            assert qw == case["qw"]
        q = _dump(hiplib, aln, 10, np.uint32, int(nq[0]) * qw).reshape(-1, qw) if nq[0] else None      # the longer p lists: q lines of qw dwords
        s_off = _dump(hiplib, aln, 6, np.uint64, n + 1)
        s_ent = _dump(hiplib, aln, 7, np.uint32, tot_p)
        c_p = _dump(hiplib, aln, 9, np.uint32, n)
        site_of_rank = np.nonzero(listed_site)[0]
        pairs_site = set()
        if case.get("long_p"):
            ks = np.diff(p_off.astype(np.int64))
            assert (ks > 30).sum() > 50, ks.max()                                   # lists beyond a 128-byte line ...
            assert qw == 64 or (ks > qw - 1).sum() > 50, ks.max()                   # ... in 128-byte lines: with an overflow line (and more)
            assert case["seed"] != 8 or (ks > qw - 1).sum() > 50, ks.max()          # (~76 per site: overflow lines at either width)
        for r in range(sites):
            t = site_of_rank[r]
            k = int(p_off[r + 1]) - int(p_off[r])
            if k == 0:
                continue
            if k <= 4:
                ents = p_ent[int(p_off[r]):int(p_off[r + 1])]
            else:
                # site r's q lines: line r = header (k | w1 << 16), qw - 2 entries, index of the first overflow line; then qw - 1 entries a line
                per = qw - 1
                hdr, ovf = int(q[r, 0]), int(q[r, per])
                assert hdr & 0xFFFF == k and (k <= per - 1 or sites <= ovf <= int(nq[0]) - (k // per))
                slot = np.arange(k) + 1
                line = np.where(slot < per, r, ovf + slot // per - 1)
                ents = q[line, slot % per]
            samp, w, mask = ents >> 5, (ents >> 4) & 1, ents & 15
            assert len(set(samp.tolist())) == samp.size
            if k > 4:
                # the w = 1 entries first, w1 of them
                w1 = hdr >> 16
                assert w1 == int(w.sum()) and (w[:w1] == 1).all() and (w[w1:] == 0).all()
            assert (M[samp, t] == mask).all() and (mask != 15).all()
            rest = np.setdiff1d(np.arange(n), np.concatenate([samp, np.nonzero(isN[:, t])[0]]))
            if rest.size:                                                 # everybody else carries the one reference base
                ref = M[rest, t]
                assert (ref == ref[0]).all() and ref[0] in (1, 2, 4, 8)
                assert (mask != ref[0]).all()
                assert (w == ((mask & ref[0]) == 0)).all()
            for s_, w_, m_ in zip(samp.tolist(), w.tolist(), mask.tolist()):
                pairs_site.add((s_, r, w_, m_))
        pairs_sample = set()
        for s_ in range(n):
            ents = s_ent[int(s_off[s_]):int(s_off[s_ + 1])]
            for e in ents.tolist():
                r_ = (e & 0x7FFFFFFF) >> 5
                assert bool(e >> 31) == (int(p_off[r_ + 1]) - int(p_off[r_]) > 4)          # flagged: the site's p list is a q line
                pairs_sample.add((s_, r_, (e >> 4) & 1, e & 15))
            assert int(c_p[s_]) == ents.size and (((ents >> 4) & 1) == 1).all()
        # the per-sample lists hold the entries that walk lists (minor_fixup_kernel): those whose mask lacks the reference base
        assert pairs_sample == {e for e in pairs_site if e[2] == 1}
        assert int(s_off[n]) == len(pairs_sample)
    # ---- the rows' N bitmaps: per site either the N plane's column (an NNL site) or nothing
    if has_T:
        T = _dump(hiplib, aln, 8, np.uint32, n * tgroups * 4).reshape(n, tgroups * 4)
        bits = np.unpackbits(T.view(np.uint8).reshape(n, -1), axis=1, bitorder="little")[:, :L].astype(bool)
        col_any = bits.any(axis=0)
        assert (bits[:, col_any] == isN[:, col_any]).all()
        assert (listed_site[col_any]).all()
        assert (isN[:, col_any].sum(axis=0) >= 2).all()
    # (small alignments: the lists are refused when they would outweigh the planes -- 128 bytes per site against 5 n / 8)
    assert has_T or nn_lists != "always" or not case.get("bitmaps")
    print("lists ok:", dict(sites=sites, lines=n_lines, p=tot_p, bitmaps=has_T), flush=True)
    aln.close()
