"""Site classes (csrc/site_classes.hip): alignments whose sites are mostly invariant run the pair kernels over the variable
sites only and complete the compared-sites counts with a one-operand pass over the invariant sites.  Results must stay
bit-identical to the oracle's pair loop (src/pairsnp.hpp:395-420) in every geometry: both encodings, plain / thresholded /
panel calls, class sizes that are not multiples of a group, no variable site at all, no invariant site at all."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
PARTIAL = np.frombuffer(b"MRWSYKVHDB", dtype=np.uint8)


def _structured(n, L, seed, mu=2e-4, p_n=0.02, p_empty=0.01, p_partial=0.0, identical=False):
    """One ancestor, per-sample substitutions at rate mu, N at rate p_n, whole columns of N at rate p_empty, optional
    partial IUPAC codes; `identical`: no substitutions at all (every site invariant or empty)."""
    rng = np.random.default_rng(seed)
    anc = BASES[rng.integers(0, 4, size=L)]
    seqs = np.tile(anc, (n, 1))
    if not identical:
        mut = rng.random((n, L)) < mu
        seqs[mut] = BASES[rng.integers(0, 4, size=int(mut.sum()))]
    seqs[rng.random((n, L)) < p_n] = ord("N")
    seqs[:, rng.random(L) < p_empty] = ord("N")
    if p_partial:
        part = rng.random((n, L)) < p_partial
        seqs[part] = PARTIAL[rng.integers(0, len(PARTIAL), size=int(part.sum()))]
    return seqs


def _check(dev, oracle, seqs, expect_classes=True, thresholds=True):
    import torch
    n, L = seqs.shape
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros_like(d)
    dev.pairsnp_dense(aln, d, nn)
    cls = aln.site_classes
    assert expect_classes is None or (cls is not None) == expect_classes, cls
    if cls is not None:
        dense, counted, minority, full = cls
        assert dense + counted + full <= L and minority <= counted + full
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    # d only (no compared-sites matrix): the counting pass is skipped
    d.zero_()
    dev.pairsnp_dense(aln, d, None)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    # a row panel against a column block (two-file geometry)
    d.zero_(); nn.zero_()
    dev.pairsnp_dense(aln, d, nn, row_begin=n // 4, row_end=n // 2, col_begin=n // 3)
    sel = (ri >= n // 4) & (ri < n // 2) & (ci >= n // 3)
    assert np.array_equal(d.cpu().numpy()[ri[sel], ci[sel]], ed[sel].astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri[sel], ci[sel]], enn[sel].astype(np.int32))
    if thresholds and len(ed):
        for q in (10, 60):
            thr = int(np.percentile(ed, q))
            d.zero_(); nn.zero_()
            dev.pairsnp_dense(aln, d, nn, dist_threshold=thr)
            keep = ed <= thr
            dh = d.cpu().numpy()
            assert np.array_equal(dh[ri[keep], ci[keep]], ed[keep].astype(np.int32))
            assert np.array_equal(nn.cpu().numpy()[ri[keep], ci[keep]], enn[keep].astype(np.int32))
            far = dh[ri[~keep], ci[~keep]].astype(np.int64)
            assert ((far > thr) | (far < 0)).all()
    _check.nw_gram = aln.nw_gram
    aln.close()
    return cls


@pytest.mark.parametrize("n,L,p_partial", [(70, 5000, 0.0), (70, 5000, 0.0005), (200, 40000, 0.0), (200, 40000, 0.0002),
                                           (131, 300001, 0.0), (131, 300001, 0.0001), (33, 129, 0.0), (700, 9000, 0.0)])
def test_site_classes_match_oracle(hiplib, oracle, n, L, p_partial):
    from tracs_amd import device as dev
    seqs = _structured(n, L, seed=n * 7 + L, p_partial=p_partial, mu=2e-4 if L > 1000 else 5e-3)
    cls = _check(dev, oracle, seqs)
    assert cls[0] + cls[2] > 0 and cls[0] < L and cls[1] > 0      # some site is variable (dense or minority), most are not


@pytest.mark.parametrize("threads", ["64", "128", "256"])
@pytest.mark.parametrize("n,L,p_partial", [(300, 20000, 0.0), (300, 20000, 0.001), (2500, 3000, 0.0005)])
def test_every_shape_of_the_classification(hiplib, oracle, monkeypatch, threads, n, L, p_partial):
    """classify_sites_kernel as workgroups of one, two and four waves (TRACS_CLASSIFY_THREADS; the default picks one by the number of
    samples): the same classes, the same distances -- incl. more samples than one step of pass 2 covers at any of the three sizes."""
    from tracs_amd import device as dev
    monkeypatch.setenv("TRACS_CLASSIFY_THREADS", threads)
    seqs = _structured(n, L, seed=n + L + int(threads), p_partial=p_partial, mu=5e-4)
    got = _check(dev, oracle, seqs)
    monkeypatch.delenv("TRACS_CLASSIFY_THREADS")
    assert got == _check(dev, oracle, seqs)                         # ... and as the default's choice classifies them


def test_no_variable_site(hiplib, oracle):
    """Identical samples (+ N): every distance is 0 and the compared-sites counts come from the invariant sites alone."""
    from tracs_amd import device as dev
    seqs = _structured(90, 7000, seed=3, identical=True)
    cls = _check(dev, oracle, seqs, thresholds=False)
    assert cls[0] == 0 and cls[2] == 0 and cls[1] > 0


def test_single_variable_site_and_ragged_classes(hiplib, oracle):
    from tracs_amd import device as dev
    seqs = _structured(150, 1000, seed=4, identical=True, p_empty=0.0)
    seqs[7, 500] = ord("A") if seqs[0, 500] != ord("A") else ord("C")
    seqs[:, 500][seqs[:, 500] == ord("N")] = seqs[0, 0]           # keep the site comparable whatever row 0 holds
    cls = _check(dev, oracle, seqs)
    assert cls[0] + cls[2] == 1


def test_dense_alignment_stays_whole(hiplib, oracle):
    """Uniformly random bases: every site is variable, the classes are not used (cost model, csrc/site_classes.hip)."""
    from tracs_amd import device as dev
    rng = np.random.default_rng(8)
    seqs = BASES[rng.integers(0, 4, size=(100, 3000))]
    seqs[rng.random(seqs.shape) < 0.02] = ord("N")
    _check(dev, oracle, seqs, expect_classes=False)


def test_repack_redecides(hiplib, oracle):
    """Packing again drops the classes of the previous contents."""
    import torch
    from tracs_amd import device as dev
    n, L = 64, 6000
    a = _structured(n, L, seed=21)
    rng = np.random.default_rng(22)
    b = BASES[rng.integers(0, 4, size=(n, L))]
    aln = dev.Alignment(n, L)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros_like(d)
    for seqs, expect in ((a, True), (b, False), (a, True)):
        aln.pack(seqs)
        dev.pairsnp_dense(aln, d, nn)
        assert (aln.site_classes is not None) == expect
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
        ri, ci = er.astype(np.int64), ec.astype(np.int64)
        assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
        assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    aln.close()


@pytest.mark.parametrize("p_partial", [0.0, 0.0004])
def test_every_class_at_once(hiplib, oracle, p_partial):
    """(consensus / general encoding) Lineage-structured columns (a fifth of the samples differ: dense), private substitutions (minority), columns without
    any N (full), columns of N (empty), N elsewhere (counted) -- in one alignment, 1 000 samples so that the list budget
    (k (cN + k) <= n^2 / 2000) separates one- and two-sample sites from the lineage sites."""
    from tracs_amd import device as dev
    n, L = 1000, 6000
    rng = np.random.default_rng(77)
    seqs = _structured(n, L, seed=78, mu=1e-4, p_n=0.0, p_empty=0.01)
    cols = rng.choice(L, size=L // 2, replace=False)
    sub = seqs[:, cols]
    sub[rng.random(sub.shape) < 0.02] = ord("N")                     # half of the columns carry N, the others none
    seqs[:, cols] = sub
    lineage = rng.choice(L, size=40, replace=False)
    members = rng.random(n) < 0.2
    for c in lineage:
        col = seqs[:, c]                                             # a view
        bases = col[col != ord("N")]
        if len(bases):
            alt = BASES[(int(np.where(BASES == bases[0])[0][0]) + 1) % 4]
            col[members & (col != ord("N"))] = alt
    if p_partial:
        part = rng.random((n, L)) < p_partial
        seqs[part] = PARTIAL[rng.integers(0, len(PARTIAL), size=int(part.sum()))]
    cls = _check(dev, oracle, seqs)
    dense, counted, minority, full = cls
    assert dense >= 30 and minority > 100 and full > (1000 if not p_partial else 100) and counted > 1000 and dense + counted + full < L


def test_general_alignment_without_dense_sites(hiplib, oracle):
    """A general alignment (partial IUPAC codes) whose variable sites all fit the lists: no site is left for the pair kernel,
    the lists and the counting pass produce everything."""
    from tracs_amd import device as dev
    n, L = 1200, 12000
    seqs = _structured(n, L, seed=31, mu=5e-5, p_n=0.01, p_empty=0.0, p_partial=3e-5)
    cls = _check(dev, oracle, seqs)
    assert cls[0] == 0 and cls[2] > 0


def _fuzz_cases():
    rng = np.random.default_rng(424242)
    out = []
    for k in range(14):
        out.append(dict(k=k, n=int(rng.choice([40, 97, 200, 333, 640, 900])), L=int(rng.choice([700, 5000, 20011, 60000])),
                        mu=float(rng.choice([0.0, 1e-4, 1e-3, 1e-2])), p_n=float(rng.choice([0.0, 0.002, 0.02, 0.2])),
                        p_empty=float(rng.choice([0.0, 0.02])), p_partial=float(rng.choice([0.0, 0.0, 1e-4, 3e-3])),
                        lineage_cols=int(rng.choice([0, 5, 60])), nrich_cols=int(rng.choice([0, 30])), seed=int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "k%d_n%d_L%d" % (c["k"], c["n"], c["L"]))
def test_site_classes_fuzz(case, hiplib, oracle):
    """Random mixtures of what decides a site's class: substitution rate, N rate (down to none: 'full' sites; up to 20 %:
    lists over budget), all-N columns, partial IUPAC codes (including codes that contain the reference base: w = 0), columns
    where a fifth of the samples differ, columns where most samples are N."""
    from tracs_amd import device as dev
    n, L = case["n"], case["L"]
    rng = np.random.default_rng(case["seed"])
    seqs = _structured(n, L, seed=case["seed"] % 100000, mu=case["mu"], p_n=case["p_n"], p_empty=case["p_empty"], p_partial=case["p_partial"])
    for c in rng.choice(L, size=min(L, case["lineage_cols"]), replace=False):
        col = seqs[:, c]
        members = rng.random(n) < 0.2
        col[members & (col != ord("N"))] = BASES[rng.integers(0, 4)]
    for c in rng.choice(L, size=min(L, case["nrich_cols"]), replace=False):
        seqs[rng.random(n) < 0.9, c] = ord("N")
    _check(dev, oracle, seqs, expect_classes=None)


# ---- the cost model's decision against the dense pass, on the regimes bench.py's `sensitivity` legs report at full size -------
REGIMES = {
    "sparse": dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01),
    "lineage": dict(mu_lineage=1e-4, mu_sample=1e-5, n_lineages=20, p_n=0.01, n_every=21),
    "divergent": dict(mu_lineage=0.0, mu_sample=1e-3, n_lineages=1, p_n=0.01),
    "clean": dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.0),
    "gappy": dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.10),
    "random": None,                                        # uniformly random bases: no site is invariant, the classes must stand down
}


@pytest.mark.parametrize("regime", sorted(REGIMES))
def test_cost_model_never_loses_to_the_dense_pass(regime, hiplib, oracle):
    """2 000 samples x 1 Mbp per regime: whatever the cost model decides (csrc/site_classes.hip: classes when
    (4 L_dense + L_counted) / 4 L < 0.92, minority lists within n^2 / 2000 entries per site) must (a) give the oracle's d and nn
    bit for bit -- checked on a 300-sample block, and over the whole matrix against the run with every site through the pair
    kernel -- and (b) not be slower than that run: a steady-state pass <= 1.1 x the dense pass, and one pass INCLUDING the
    once-per-pack work <= 1.1 x the dense run's.  The reference's cost is the same on all of them (src/pairsnp.hpp:395-420)."""
    import torch
    from tracs_amd import device as dev, synth
    n, L = 2000, 1_000_000
    aln = dev.Alignment(n, L)
    if REGIMES[regime] is None:
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device="cuda")
        for s0 in range(0, n, 100):
            idx = torch.randint(0, 4, (100, L), generator=g, device="cuda")
            idx[torch.rand((100, L), generator=g, device="cuda") < 0.01] = 4
            aln.pack(lut[idx], first=s0)
        head = None
    else:
        synth.pack_synthetic_device(aln, seed=31, **REGIMES[regime])
        head = synth.first_samples_host(n, L, 31, 300, **REGIMES[regime])
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros_like(d)

    def run():
        first = 1e9
        for _ in range(2):                                    # (twice: the process's first launch of a kernel loads its code)
            aln.mark_packed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.pairsnp_dense(aln, d, nn)
            torch.cuda.synchronize()
            first = min(first, time.perf_counter() - t0)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            dev.pairsnp_dense(aln, d, nn)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return first, best
    try:
        first_on, pass_on = run()
        classes = aln.site_classes
        d_on, nn_on = d.clone(), nn.clone()
        hiplib.tracs_debug_force_site_classes(0)
        aln.mark_packed()
        d.zero_(); nn.zero_()
        first_off, pass_off = run()
        assert aln.site_classes is None
    finally:
        hiplib.tracs_debug_force_site_classes(-2)
    assert bool(torch.equal(torch.triu(d_on, 1), torch.triu(d, 1))) and bool(torch.equal(torch.triu(nn_on, 1), torch.triu(nn, 1)))
    if head is not None:
        er, ec, ed, enn = oracle.pairsnp_arrays(head, n_threads=16)
        ri, ci = er.astype(np.int64), ec.astype(np.int64)
        assert np.array_equal(d_on[:300, :300].cpu().numpy()[ri, ci], ed.astype(np.int32))
        assert np.array_equal(nn_on[:300, :300].cpu().numpy()[ri, ci], enn.astype(np.int32))
    assert (classes is None) == (regime == "random"), (regime, classes)
    print("%s: classes %s  pass %.2f ms vs dense %.2f ms; first pass %.2f vs %.2f ms" %
          (regime, classes, pass_on * 1e3, pass_off * 1e3, first_on * 1e3, first_off * 1e3))
    if classes is not None:
        assert pass_on <= 1.1 * pass_off, (regime, pass_on, pass_off)
        assert first_on <= 1.1 * first_off, (regime, first_on, first_off)
    aln.close()


def test_row_hint_builds_lists_for_those_rows_only(hiplib, oracle):
    """tracs_alignment_hint_rows (multi-GPU ranks): the per-sample lists of the site classes exist for the promised rows only --
    results on those rows are the oracle's, a call for other rows fails, lifting the promise rebuilds."""
    import torch
    from tracs_amd import device as dev
    n, L = 700, 9000
    seqs = _structured(n, L, seed=91, mu=3e-4, p_n=0.01)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros_like(d)
    try:
        hiplib.tracs_debug_force_site_classes(1)
        ranges = [(64, 192), (500, 700)]
        aln.hint_rows(ranges)
        for r0, r1 in ranges:
            dev.pairsnp_dense(aln, d, nn, row_begin=r0, row_end=r1)
        assert aln.site_classes is not None and aln.site_classes[2] > 0
        sel = ((ri >= 64) & (ri < 192)) | (ri >= 500)
        assert np.array_equal(d.cpu().numpy()[ri[sel], ci[sel]], ed[sel].astype(np.int32))
        assert np.array_equal(nn.cpu().numpy()[ri[sel], ci[sel]], enn[sel].astype(np.int32))
        with pytest.raises(RuntimeError, match="hint_rows"):
            dev.pairsnp_dense(aln, d, nn, row_begin=0, row_end=64)
        with pytest.raises(RuntimeError, match="hint_rows"):
            dev.pairsnp_dense(aln, d, nn)
        aln.hint_rows([])
        d.zero_(); nn.zero_()
        dev.pairsnp_dense(aln, d, nn)
        assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32)) and np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    finally:
        hiplib.tracs_debug_force_site_classes(-2)
    aln.close()


LONG_LISTS_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from tracs_amd import device as dev, synth
n, L = 1500, 6000
seqs = synth.alignment(n, L, seed=77, mu_lineage=1e-3, mu_sample=2e-4, n_lineages=7, p_n=0.002)
rng = np.random.default_rng(78)
gappy = rng.choice(L, L // 10, replace=False)                      # a tenth of the sites with many N samples (the lists' cap: n L / 8 entries)
mask = rng.random((n, gappy.size)) < %(p_n)g
sub = seqs[:, gappy]; sub[mask] = ord("N"); seqs[:, gappy] = sub
aln = dev.Alignment(n, L)
aln.pack(seqs)
er, ec, ed, enn = O.pairsnp_arrays(seqs, n_threads=16)
ri, ci = er.astype(np.int64), ec.astype(np.int64)
d = torch.zeros((n, n), dtype=torch.int32, device="cuda"); nn = torch.zeros_like(d)
dev.pairsnp_dense(aln, d, nn)
src = aln.count_source
assert aln.site_classes is not None and src is not None and src[2] > L // 2, (aln.site_classes, src)      # N co-occurrences from lists
assert aln.list_stats["nn_visits"] > (L // 10) * (%(p_n)g * n) ** 2 / 2, aln.list_stats              # ... the long ones too
assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
# row panels at both ends and in the middle
for r0, r1 in ((0, 64), (700, 830), (1400, n)):
    dp = torch.zeros((r1 - r0, n), dtype=torch.int32, device="cuda"); npn = torch.zeros_like(dp)
    dev.pairsnp_dense(aln, dp, npn, row_begin=r0, row_end=r1, base_row=r0)
    sel = (ri >= r0) & (ri < r1)
    assert np.array_equal(dp.cpu().numpy()[ri[sel] - r0, ci[sel]], ed[sel].astype(np.int32)), (r0, r1)
    assert np.array_equal(npn.cpu().numpy()[ri[sel] - r0, ci[sel]], enn[sel].astype(np.int32)), (r0, r1)
print("LISTS", aln.list_stats)
aln.close()
'''


@pytest.mark.parametrize("env", [{"TRACS_NN_LIST_K": "1"}], ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()))
@pytest.mark.parametrize("p_n", [0.15, 0.05])
def test_long_n_lists(hiplib, oracle, env, p_n):
    """N lists of ~ 225 (75) samples at 1 500 samples -- two 124-byte lines per list at the higher rate, so the walks of
    nn_rows_kernel and minor_fixup_kernel go on into overflow lines, and the per-site pass takes its groups in several pieces --
    forced onto the lists whatever the cost model says; every pair against the oracle, whole matrix and row panels."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", LONG_LISTS_CHILD % {"root": root, "p_n": p_n}], capture_output=True, text=True,
                         env=dict(os.environ, **env), timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.parametrize("n,p_partial", [(65534, 0.0), (65600, 0.0), (65600, 0.0003)])
def test_lists_at_the_16_bit_boundary(hiplib, oracle, n, p_partial):
    """65 534 / 65 600 samples: two column chunks per row, and -- beyond 65 536 samples -- the per-site pass takes every group in
    pieces (its LDS staging holds 16-bit offsets from a piece's first sample); lists of a few N samples among 65 000: long runs of
    skip bytes.  With partial codes (~20 listed samples per site, ~2 500 per group) the p lists are p_lists_kernel's: its queue of
    flagged samples in chunks of 4 096, sample numbers beyond 16 bits in the q lines.  Row panels against the oracle on a subset of
    the columns (the panel rows + 1 500 random samples)."""
    import torch
    from tracs_amd import device as dev, synth
    L = 512
    seqs = synth.alignment(n, L, seed=123, mu_lineage=2e-3, mu_sample=3e-4, n_lineages=9, p_n=0.004, p_partial=p_partial)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    rng = np.random.default_rng(5)
    others = np.sort(rng.choice(n, 1500, replace=False))
    for r0, r1 in ((10, 26), (32760, 32776), (40000, 40016), (n - 16, n)):
        dp = torch.zeros((r1 - r0, n), dtype=torch.int32, device="cuda")
        npn = torch.zeros_like(dp)
        dev.pairsnp_dense(aln, dp, npn, row_begin=r0, row_end=r1, base_row=r0)
        assert aln.site_classes is not None and aln.count_source[2] > 0, (aln.site_classes, aln.count_source)
        rows = np.arange(r0, r1)
        sub = np.unique(np.concatenate([rows, others]))
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs[sub], n_threads=16)
        gi, gj = sub[er.astype(np.int64)], sub[ec.astype(np.int64)]             # (i < j in the subset's order = the global order)
        sel = (gi >= r0) & (gi < r1)
        dh, nh = dp.cpu().numpy(), npn.cpu().numpy()
        assert np.array_equal(dh[gi[sel] - r0, gj[sel]], ed[sel].astype(np.int32)), (r0, r1)
        assert np.array_equal(nh[gi[sel] - r0, gj[sel]], enn[sel].astype(np.int32)), (r0, r1)
    aln.close()
