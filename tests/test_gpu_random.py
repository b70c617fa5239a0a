"""Randomised geometry sweep of the dense pair entry points against the oracle: sample counts around every tile edge
(64 / 128), alignment lengths around group (128-site) and stage boundaries and across the two-pass threshold switch, row
panels and column starts as the multi-GPU driver and the two-file mode use them, with and without ncomp, both encodings,
plain and thresholded.  Seeded: the same 40 cases every run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20241022)
    n_pool = [2, 3, 63, 64, 65, 127, 128, 129, 191, 200, 257, 300, 511, 640]
    l_pool = [1, 31, 127, 128, 129, 255, 256, 257, 1000, 4096, 16383, 16384, 16385, 40001, 99991, 131072]
    out = []
    for k in range(40):
        n = int(rng.choice(n_pool))
        L = int(rng.choice(l_pool))
        if k % 8 == 7:
            n, L = int(rng.integers(300, 700)), int(rng.integers(20000, 60000))
        r0 = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
        r1 = int(rng.integers(r0 + 1, n + 1)) if rng.random() < 0.5 else n
        cb = int(rng.integers(0, n)) if rng.random() < 0.35 else 0
        out.append(dict(k=k, n=n, L=L, r0=r0, r1=r1, cb=cb, partial=bool(rng.random() < 0.4), with_nn=bool(rng.random() < 0.75),
                        thr=(None if rng.random() < 0.5 else int(rng.integers(0, 60))), seed=int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "k%d_n%d_L%d" % (c["k"], c["n"], c["L"]))
def test_dense_entry_points_random_geometry(case, hiplib, oracle):
    import torch
    from tracs_amd import device as dev
    from tracs_amd import synth
    n, L = case["n"], case["L"]
    seqs = synth.alignment(n, L, seed=case["seed"], mu_lineage=2e-3, mu_sample=3e-4, n_lineages=5, p_n=0.02,
                           p_partial=0.01 if case["partial"] else 0.0)
    seqs = seqs[np.argsort(np.arange(n) % 5, kind="stable")]                # lineages contiguous: whole tiles far apart
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    fill = -7
    d = torch.full((n, n + 3), fill, dtype=torch.int32, device="cuda")       # ld > n on purpose
    nn = torch.full((n, n + 3), fill, dtype=torch.int32, device="cuda") if case["with_nn"] else None
    dev.pairsnp_dense(aln, d, nn, row_begin=case["r0"], row_end=case["r1"], col_begin=case["cb"], dist_threshold=case["thr"])
    torch.cuda.synchronize()
    assert aln.encoding == ("general" if case["partial"] and _has_partial(seqs) else "consensus")
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
    want_d = np.full((n, n), -1, np.int64)
    want_n = np.full((n, n), -1, np.int64)
    want_d[er.astype(np.int64), ec.astype(np.int64)] = ed.astype(np.int64)
    want_n[er.astype(np.int64), ec.astype(np.int64)] = enn.astype(np.int64)
    got_d = d.cpu().numpy()[:, :n].astype(np.int64)
    got_n = nn.cpu().numpy()[:, :n].astype(np.int64) if nn is not None else None
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    inside = (ii >= case["r0"]) & (ii < case["r1"]) & (jj > ii) & (jj >= case["cb"])
    # cells outside the requested set are never touched (padding columns included)
    assert (got_d[~inside] == fill).all() and (d.cpu().numpy()[:, n:] == fill).all()
    if got_n is not None:
        assert (got_n[~inside] == fill).all()
    thr = case["thr"]
    if thr is None:
        assert np.array_equal(got_d[inside], want_d[inside])
        if got_n is not None:
            assert np.array_equal(got_n[inside], want_n[inside])
    else:
        keep = inside & (want_d <= thr)
        far = inside & (want_d > thr)
        assert np.array_equal(got_d[keep], want_d[keep])
        if got_n is not None:
            assert np.array_equal(got_n[keep], want_n[keep])
        u = got_d[far] & 0xFFFFFFFF                                         # exact, or marked with bit 31: never <= thr
        assert ((u > thr) | (u >= 0x80000000)).all()
        if aln.site_classes is None and aln.encoding == "consensus":       # (otherwise a dead cell may hold any value > thr:
            assert ((u == (want_d[far] & 0xFFFFFFFF)) | (u >= 0x80000000)).all()     # a lower bound + the terms added afterwards)
        rows, cols, dd, nc = dev.coo_from_dense(d, nn, n, dist_threshold=thr, row_begin=case["r0"], row_end=case["r1"],
                                                col_begin=case["cb"])
        sel = keep[er.astype(np.int64), ec.astype(np.int64)]
        assert np.array_equal(rows.cpu().numpy().astype(np.int64), er[sel].astype(np.int64))
        assert np.array_equal(cols.cpu().numpy().astype(np.int64), ec[sel].astype(np.int64))
        assert np.array_equal(dd.cpu().numpy().astype(np.int64), ed[sel].astype(np.int64))
    aln.close()


def _has_partial(seqs):
    ok = np.zeros(256, bool)
    for ch in b"ACGTacgt":
        ok[ch] = True
    full = ~ok
    for ch in b"MRWSYKVHDBmrwsykvhdb":
        full[ch] = False
    return bool((~ok & ~full)[seqs].any())


def test_matrix_core_kernel_extremes(hiplib, oracle):
    """The consensus pass accumulates S = 4 * matches - nn and nn in fp32 matrix-core accumulators.  Push both to their extremes
    over a range longer than one exact chunk (2^22 sites, so the host must split it): samples that differ at EVERY site
    (S = -nn = -L), that agree everywhere (S = 3 L), that share no compared site, and patterned ones."""
    import torch
    from tracs_amd import device as dev
    L = (1 << 22) + 70003
    rng = np.random.default_rng(8)
    base = np.frombuffer(b"ACGT", np.uint8)
    rows = [np.full(L, ord("A"), np.uint8), np.full(L, ord("C"), np.uint8), np.full(L, ord("G"), np.uint8), np.full(L, ord("T"), np.uint8),
            np.full(L, ord("N"), np.uint8), np.full(L, ord("A"), np.uint8), base[np.arange(L) & 3].copy(), base[(np.arange(L) >> 5) & 3].copy(),
            base[rng.integers(0, 4, L)]]
    rows[7][::3] = ord("-")
    rows[8][rng.random(L) < 0.3] = ord("n")
    seqs = np.array(rows)
    n = len(seqs)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    assert aln.encoding == "consensus" and aln.kernel == "mfma"
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    assert ed[(ri == 0) & (ci == 1)][0] == L and ed[(ri == 0) & (ci == 5)][0] == 0 and enn[(ri == 0) & (ci == 4)][0] == 0
    aln.close()


@pytest.mark.parametrize("partial", [False, True], ids=["consensus", "general"])
def test_panel_that_runs_past_the_sample_padding(partial, hiplib, oracle):
    """n equal to the sample padding (2048) and an unaligned row panel: the last tile's rows start inside the alignment and
    end beyond the padded sample count, so its staging reads run into the next plane row (garbage for rows >= n, which must
    stay masked) or, for the very last row, into the zeroed tail."""
    import torch
    from tracs_amd import device as dev
    from tracs_amd import synth
    n, L, r0, r1 = 2048, 700, 1957, 2048
    seqs = synth.alignment(n, L, seed=77, mu_lineage=5e-3, mu_sample=2e-3, n_lineages=9, p_n=0.03, p_partial=0.01 if partial else 0.0)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.full((n, n), -3, dtype=torch.int32, device="cuda")
    nn = torch.full((n, n), -3, dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn, row_begin=r0, row_end=r1)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs[r0:], n_threads=8)
    ri, ci = er.astype(np.int64) + r0, ec.astype(np.int64) + r0
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32)) and np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    assert int((d[:r0] != -3).sum().item()) == 0
    aln.close()
