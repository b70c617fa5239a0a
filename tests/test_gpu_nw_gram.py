"""The second form of the site classes (csrc/site_classes.hip, nw_gram): the minority sites' N x listed terms from two one-plane
passes on the matrix cores (U U^T - n n^T) instead of walks of the sites' N lists -- what alignments masked by coverage (`tracs align`
writes N wherever a sample's depth is below its thresholds, tracs/align.py:599-613: tens of per cent of every sample) need, where
k (cN + k) would send every site through the pair kernel.  Results bit-identical to the oracle's pair loop (src/pairsnp.hpp:395-420)
in every geometry _check exercises (plain / d only / panel against a column block / thresholded), forced on ordinary alignments
and chosen by the cost model on masked ones."""
import numpy as np
import pytest

from test_gpu_site_classes import BASES, PARTIAL, _check, _structured

pytestmark = pytest.mark.gpu


def _masked(n, L, seed, frac=0.3, mean_len=400, p_partial=0.0, mu=2e-4, lineages=1):
    """per-sample coverage gaps: every sample N over `frac` of the sites in runs of its own (+ partial codes, + lineages)"""
    rng = np.random.default_rng(seed)
    anc = BASES[rng.integers(0, 4, size=L)]
    founders = [anc.copy() for _ in range(lineages)]
    for f in founders[1:]:
        m = rng.random(L) < 5 * mu
        f[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
    seqs = np.stack([founders[s % lineages] for s in range(n)])
    mut = rng.random((n, L)) < mu
    seqs[mut] = BASES[rng.integers(0, 4, size=int(mut.sum()))]
    if p_partial:
        part = rng.random((n, L)) < p_partial
        seqs[part] = PARTIAL[rng.integers(0, len(PARTIAL), size=int(part.sum()))]
    k = max(1, int(round(-np.log(1.0 - frac) * L / mean_len)))
    for s in range(n):
        st = rng.integers(0, L, size=k)
        ln = rng.geometric(1.0 / mean_len, size=k)
        edge = np.zeros(L + 1, np.int32)
        np.add.at(edge, st, 1)
        np.add.at(edge, np.minimum(st + ln, L), -1)
        seqs[s, np.cumsum(edge)[:L] > 0] = ord("N")
    return seqs


@pytest.mark.parametrize("rows", ["0", "1"], ids=["u-pass", "ns-rows"])
@pytest.mark.parametrize("n,L,p_partial", [(70, 5000, 0.0), (70, 5000, 0.0005), (200, 40000, 0.0002), (131, 300001, 0.0001), (33, 129, 0.0),
                                           (700, 9000, 0.0), (260, 20000, 0.01)])
def test_forced_on_ordinary_alignments(hiplib, oracle, monkeypatch, n, L, p_partial, rows):
    """both ways the second form gets its N x listed terms: the U pass on the matrix cores, and the rows of the site-major N matrix
    summed per listed sample inside the fix-up (minor_fixup_kernel<NSROWS>)"""
    from tracs_amd import device as dev
    monkeypatch.setenv("TRACS_NW_GRAM", "1")
    monkeypatch.setenv("TRACS_NW_ROWS", rows)
    seqs = _structured(n, L, seed=n * 7 + L, p_partial=p_partial, mu=2e-4 if L > 1000 else 5e-3)
    cls = _check(dev, oracle, seqs, expect_classes=None)      # (a small general alignment may still prefer the VALU kernel: cost model)
    assert cls is None or (cls[2] > 0 and _check.nw_gram)
    assert cls is not None or p_partial >= 0.01


@pytest.mark.parametrize("n,L,p_partial,lineages", [(200, 20000, 0.01, 2), (320, 30011, 0.005, 1), (200, 20000, 0.0, 2), (1100, 6000, 0.004, 3),
                                                    (515, 12345, 0.0, 1)])
def test_chosen_for_alignments_masked_by_coverage(hiplib, oracle, monkeypatch, n, L, p_partial, lineages):
    """30 % N per sample in runs of its own: the list form is refused (every site dense), the matrix-core form takes over"""
    from tracs_amd import device as dev
    monkeypatch.delenv("TRACS_NW_GRAM", raising=False)
    monkeypatch.delenv("TRACS_NW_ROWS", raising=False)
    seqs = _masked(n, L, seed=n + L, p_partial=p_partial, lineages=lineages)
    cls = _check(dev, oracle, seqs, expect_classes=None)
    print(n, L, p_partial, cls, _check.nw_gram)
    # (cost model: without partial codes few sites are variable and the list form stays; a small alignment with partial codes may
    # prefer the VALU kernel over any classes; from ~1 000 samples on the matrix-core form is what runs)
    if p_partial and n >= 1000:
        assert cls is not None and cls[2] > 0.8 * L and _check.nw_gram, cls
    # and never when switched off: the same results from the form without it
    monkeypatch.setenv("TRACS_NW_GRAM", "0")
    _check(dev, oracle, seqs, expect_classes=None)
    assert not _check.nw_gram


@pytest.mark.parametrize("rows", ["0", "1"], ids=["u-pass", "ns-rows"])
@pytest.mark.parametrize("n,p_partial", [(40000, 0.0003), (65600, 0.0)])
def test_two_column_chunks(hiplib, oracle, monkeypatch, n, p_partial, rows):
    """beyond 36 800 samples a row of the pair matrix is two column chunks of the fix-up (the NS rows' words are cut with them), beyond
    65 536 the per-site pass takes every group in pieces: row panels against the oracle on a subset of the columns"""
    import torch
    from tracs_amd import device as dev, synth
    monkeypatch.setenv("TRACS_NW_GRAM", "1")
    monkeypatch.setenv("TRACS_NW_ROWS", rows)
    L = 512
    seqs = synth.alignment(n, L, seed=321, mu_lineage=2e-3, mu_sample=3e-4, n_lineages=9, p_n=0.05, p_partial=p_partial)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    rng = np.random.default_rng(6)
    others = np.sort(rng.choice(n, 1200, replace=False))
    for r0, r1 in ((10, 26), (36790, 36806), (n - 16, n)):
        dp = torch.zeros((r1 - r0, n), dtype=torch.int32, device="cuda")
        npn = torch.zeros_like(dp)
        dev.pairsnp_dense(aln, dp, npn, row_begin=r0, row_end=r1, base_row=r0)
        assert aln.site_classes is not None and aln.nw_form == ("ns-rows" if rows == "1" else "u-pass"), (aln.site_classes, aln.nw_form)
        sub = np.unique(np.concatenate([np.arange(r0, r1), others]))
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs[sub], n_threads=16)
        gi, gj = sub[er.astype(np.int64)], sub[ec.astype(np.int64)]
        sel = (gi >= r0) & (gi < r1)
        dh, nh = dp.cpu().numpy(), npn.cpu().numpy()
        assert np.array_equal(dh[gi[sel] - r0, gj[sel]], ed[sel].astype(np.int32)), (r0, r1)
        assert np.array_equal(nh[gi[sel] - r0, gj[sel]], enn[sel].astype(np.int32)), (r0, r1)
    aln.close()
