"""Recombination filter (src/pairsnp.hpp:223-318).  PARITY UNPINNED: the reference needs Boost's binomial CDF
and its golden input long_filt.aln is not in the tree, so the oracle is checked against first principles
(scipy's binomial CDF, an independent Python restatement) and the GPU against the oracle."""
import os

import numpy as np
import pytest


def _py_filter(pos, L):
    """Independent restatement of filter_recomb with scipy's binomial CDF."""
    from scipy.stats import binom
    d = len(pos)
    if d <= 1:
        return d
    p = d / L
    thr = 0.05 / d
    w = max(min(int(1.0 / p / 2.0 + 1), 5000), 50)
    kept = 0
    pos = np.asarray(pos)
    for i in pos:
        left, right = max(0, i - w), min(L, i + w + 1)
        inside = pos[(pos >= left) & (pos < right)]
        if len(inside) > 1:
            n = int(inside[-1] - inside[0] + 1)
            if 1.0 - binom.cdf(len(inside), n, p) >= thr:
                kept += 1
        else:
            kept += 1
    return kept


def test_oracle_binomial_cdf_vs_scipy(oracle):
    from scipy.stats import binom
    rng = np.random.default_rng(1)
    for _ in range(2000):
        n = int(rng.integers(1, 10002))
        p = float(rng.choice([1e-6, 1e-4, 2e-3, 0.05, 0.3]))
        k = int(rng.integers(0, min(n, 80) + 1))
        assert abs(oracle.binomial_cdf(n, p, k) - binom.cdf(k, n, p)) < 1e-9


def test_oracle_filter_vs_python_restatement(oracle):
    rng = np.random.default_rng(2)
    for L, d, block in ((100000, 30, 0), (100000, 30, 25), (2000000, 400, 150), (5000, 2, 0), (5000, 1, 0), (5000, 0, 0)):
        pos = set(rng.choice(L, size=d, replace=False).tolist())
        if block:
            start = int(rng.integers(0, L - 600))
            pos |= set((start + rng.choice(500, size=block, replace=False)).tolist())
        pos = sorted(pos)
        assert oracle.filter_recomb_positions(pos, L) == _py_filter(pos, L)


def test_binomial_tail_grows_with_the_span(oracle):
    """What the GPU's threshold rows rest on (csrc/filter_lists.hip: per d, the smallest span that survives with `count` SNPs in the
    window): 1 - BinomCDF(count; n, p) >= 0.05 / d, once true, stays true for every longer span -- checked with the oracle's CDF over
    the whole range a window can have (count < n <= 2 w + 1) for distances from a handful to tens of thousands."""
    for L, d in ((5000000, 980), (5000000, 21000), (5000000, 2), (120000, 37), (9000, 600), (1000000, 65536)):
        p, thr = d / L, 0.05 / d
        w = max(min(int(1.0 / p / 2.0 + 1), 5000), 50)
        nmax = min(L, 2 * w + 1)
        for k in (2, 3, 5, 9, 17, 40, 63):
            if k + 1 > nmax:
                continue
            ns = np.unique(np.concatenate([np.arange(k + 1, min(nmax, k + 400) + 1), np.linspace(k + 1, nmax, 300).astype(int)]))
            keep = np.array([1.0 - oracle.binomial_cdf(int(n), p, k) >= thr for n in ns])
            first = int(np.argmax(keep)) if keep.any() else len(keep)
            assert keep[first:].all() and not keep[:first].any(), (L, d, k)


def test_oracle_filter_removes_a_planted_block(oracle):
    from tracs_amd import synth
    L = 150000
    seqs = synth.alignment(5, L, seed=3, mu_lineage=3e-4, mu_sample=1e-4, p_n=0.0)
    lut = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"CGTA"):
        lut[a] = b
    seqs[1, 60000:60300] = lut[seqs[1, 60000:60300]]          # 300 consecutive substitutions in sample 1
    r, c, d, _ = oracle.pairsnp_arrays(seqs)
    f = oracle.filter_recomb_pairs(seqs, r, c, 2)
    for ri, ci, di, fi in zip(r, c, d, f):
        assert fi <= di
        if 1 in (ri, ci):
            assert di >= 300 and fi <= di - 290          # the block is filtered out
    out = oracle.pairsnp([_write(seqs)], 1, 2147483647, True)
    assert out[4] == f.tolist()


def _write(seqs):
    import tempfile
    from tracs_amd import synth
    p = os.path.join(tempfile.mkdtemp(), "f.fa")
    synth.write_fasta(p, seqs)
    return p


@pytest.fixture(params=["wave_per_pair", "pair_per_lane"])
def extract_kernel(request, monkeypatch):
    """Both extraction kernels: the library picks by the number of emitted pairs (TRACS_FILTER_LANES_MIN overrides)."""
    monkeypatch.setenv("TRACS_FILTER_LANES_MIN", "0" if request.param == "pair_per_lane" else "1000000000000")
    return request.param


# every way tracs_filter_recomb_pairs can take: lists + table (the default: the merge-path kernel), lists with the tail summed per
# SNP, the scan of the planes for every pair, lists whose LDS capacity sends pairs to the long-list kernels (tiled: the default;
# sequential), and the binary-search kernel
ROUTES = {"lists": {}, "lists_no_table": {"TRACS_FILTER_TABLE": "0"}, "scan": {"TRACS_FILTER_LISTS": "0"},
          "scan_batches": {"TRACS_FILTER_LISTS": "0", "TRACS_FILTER_SCAN_MAXPOS": "20000"},
          "scan_batches_lanes": {"TRACS_FILTER_LISTS": "0", "TRACS_FILTER_SCAN_MAXPOS": "50000", "TRACS_FILTER_LANES_MIN": "0"},
          "lists_cap64_batches": {"TRACS_FILTER_CAP": "64", "TRACS_FILTER_SCAN_MAXPOS": "3000"},
          "lists_cap64": {"TRACS_FILTER_CAP": "64"}, "lists_cap64_sequential_long_kernel": {"TRACS_FILTER_CAP": "64", "TRACS_FILTER_LONG": "seq"}, "lists_search_kernel": {"TRACS_FILTER_KERNEL": "1"},
          "lists_search_kernel_cap128": {"TRACS_FILTER_KERNEL": "1", "TRACS_FILTER_CAP": "128"}}


@pytest.fixture(params=sorted(ROUTES))
def route(request, monkeypatch):
    for k, v in ROUTES[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


def _blocky(oracle, n, L, seed, **kw):
    """an alignment with recombination-like blocks (runs of substitutions, one touching each end of the alignment)"""
    from tracs_amd import synth
    seqs = synth.alignment(n, L, seed=seed, **kw)
    lut = np.zeros(256, np.uint8) + ord("N")
    for a, b in zip(b"ACGT", b"CGTA"):
        lut[a] = b
    rng = np.random.default_rng(seed)
    for s in rng.choice(n, size=max(2, n // 4), replace=False):
        a = int(rng.integers(0, L - 400))
        w = int(rng.integers(20, 400))
        keep = rng.random(w) < rng.choice([0.15, 0.5, 1.0])
        seg = seqs[s, a:a + w]
        seqs[s, a:a + w] = np.where(keep & (seg != ord("N")), lut[seg], seg)
    seqs[1, :120] = lut[seqs[1, :120]]
    seqs[2, L - 90:] = lut[seqs[2, L - 90:]]
    return seqs


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(24, 120000, dict(mu_lineage=4e-4, mu_sample=1e-4, p_n=0.01, p_partial=0.002)),
                                   (40, 70001, dict(mu_lineage=1e-3, mu_sample=3e-4, p_n=0.02, p_partial=0.01, p_lower=0.1)),
                                   (33, 50000, dict(mu_lineage=0.0, mu_sample=2e-3, p_n=0.1)),
                                   (12, 9000, dict(mu_lineage=2e-2, mu_sample=5e-2, p_n=0.05, p_partial=0.02)),
                                   (70, 30011, dict(mu_lineage=0.0, mu_sample=1e-4, p_n=0.0))])
def test_gpu_filter_pairs_matches_oracle(oracle, hiplib, route, shape):
    """tracs_filter_recomb_pairs (every route) = the oracle's filter_recomb on every pair, thresholded or not"""
    import torch
    from tracs_amd import device as dev
    n, L, kw = shape
    seqs = _blocky(oracle, n, L, seed=n + L, **kw)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    r, c, d, _ = oracle.pairsnp_arrays(seqs)
    ef = oracle.filter_recomb_pairs(seqs, r, c, 4)
    assert (ef < d).any()
    for sel in (np.ones(len(r), bool), d <= np.median(d), np.arange(len(r)) % 7 == 3):
        rows = torch.from_numpy(r[sel].astype(np.int32)).cuda()
        cols = torch.from_numpy(c[sel].astype(np.int32)).cuda()
        dd = torch.from_numpy(d[sel].astype(np.int32)).cuda()
        got = dev.filter_recomb_pairs(aln, rows, cols, dd).cpu().numpy()
        assert np.array_equal(got, ef[sel].astype(np.int32)), (route, np.where(got != ef[sel])[0][:10])
    info = dev.filter_index_info(aln)
    assert info is not None and (not info["lists"] if route.startswith("scan") else info["lists"] or kw["mu_sample"] > 1e-2)
    # a wrong distance is caught, never filtered silently
    bad = torch.from_numpy((d + 1).astype(np.int32)).cuda()
    with pytest.raises(RuntimeError, match="does not match the distance"):
        dev.filter_recomb_pairs(aln, torch.from_numpy(r.astype(np.int32)).cuda(), torch.from_numpy(c.astype(np.int32)).cuda(), bad)


@pytest.mark.gpu
def test_gpu_filter_index_follows_the_planes(oracle, hiplib):
    """the departure lists are rebuilt when the handle is packed again"""
    import torch
    from tracs_amd import device as dev
    n, L = 16, 40000
    aln = dev.Alignment(n, L)
    for seed in (1, 2):
        seqs = _blocky(oracle, n, L, seed=seed, mu_lineage=1e-3, mu_sample=2e-4, p_n=0.03)
        aln.pack(seqs)
        r, c, d, _ = oracle.pairsnp_arrays(seqs)
        got = dev.filter_recomb_pairs(aln, *(torch.from_numpy(x.astype(np.int32)).cuda() for x in (r, c, d))).cpu().numpy()
        assert np.array_equal(got, oracle.filter_recomb_pairs(seqs, r, c, 4).astype(np.int32))


@pytest.mark.gpu
def test_gpu_filter_matches_oracle(oracle, hiplib, tmp_path, extract_kernel):
    import torch  # noqa: F401
    from tracs_amd import api, synth
    L = 120000
    seqs = synth.alignment(24, L, seed=8, mu_lineage=4e-4, mu_sample=1e-4, p_n=0.01, p_partial=0.002)
    lut = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"CGTAN"):
        lut[a] = b
    seqs[3, 1000:1200] = lut[seqs[3, 1000:1200]]
    seqs[7, L - 150:] = lut[seqs[7, L - 150:]]                   # block touching the end of the alignment
    fa = os.path.join(str(tmp_path), "f.fa")
    synth.write_fasta(fa, seqs, width=70)
    med = int(np.median(oracle.pairsnp_arrays(seqs)[2]))
    for dist in (2147483647, med):
        r, c, d, names, filt, nn = api.pairsnp_arrays([fa], 1, dist, True)
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs, dist=dist)
        assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(d, ed) and np.array_equal(nn, enn)
        ef = oracle.filter_recomb_pairs(seqs, er, ec, 4)
        assert np.array_equal(filt, ef), np.where(filt != ef)
        assert len(d) > 10 and (filt <= d).all() and (filt < d).any()
    out = api.pairsnp(fasta=[fa], n_threads=1, dist=med, filter=True)
    assert out[4] == ef.tolist() and all(isinstance(v, int) for v in out[4])


@pytest.mark.gpu
def test_gpu_filter_positions_are_the_snp_sites(oracle, hiplib, extract_kernel):
    import torch
    from tracs_amd import device as dev
    from tracs_amd import synth
    L, n = 70001, 9
    seqs = synth.alignment(n, L, seed=5, mu_lineage=1e-3, mu_sample=3e-4, p_n=0.02, p_partial=0.01, p_lower=0.1)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    r, c, d, _ = oracle.pairsnp_arrays(seqs)
    rows = torch.from_numpy(r.astype(np.int32)).cuda()
    cols = torch.from_numpy(c.astype(np.int32)).cuda()
    dd = torch.from_numpy(d.astype(np.int32)).cuda()
    filt, found, pos, off = dev.filter_recomb_device(aln, rows, cols, dd)
    assert np.array_equal(found.cpu().numpy(), d.astype(np.int32))
    m = oracle._MASK[seqs]
    pos, off = pos.cpu().numpy(), off.cpu().numpy()
    for t in range(len(r)):
        sites = np.nonzero((m[r[t]] & m[c[t]]) == 0)[0]
        assert np.array_equal(pos[off[t]:off[t + 1]], sites.astype(np.int32))
