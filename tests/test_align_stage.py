"""The post-pileup stage of `tracs align` on the device: allele counts -> find_dirichlet_priors ->
calculate_posteriors -> coverage rules -> IUPAC code (tracs/align.py:536-622), and packing the codes straight into the
alignment planes without the FASTA round trip.  The numpy restatement below follows the reference lines cited and uses
the oracle for the posterior filter; the GPU chain must reproduce its sequence letter for letter, and pairsnp on the
packed codes must equal pairsnp on the FASTA made from those letters."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# np.packbits(bitorder="little") index -> letter (tracs/align.py:285-323)
_LUT = np.frombuffer(b"XACMGRSVTWYHKDBN", dtype=np.uint8)


def _reference_sequence(oracle, counts, alphas, keep, thr, min_cov, band):
    post = oracle.calculate_posteriors(counts.astype(np.float64), alphas, keep, thr)      # align.py:575-577
    rs = counts.sum(1)
    if band is not None:                                                                   # :599-612
        post[(rs <= band[1]) & (rs >= band[0])] = 1
    post[rs < min_cov] = 1                                                                 # :613
    idx = np.packbits(post > 0, axis=1, bitorder="little").flatten()                      # :616-622
    return _LUT[idx]


def test_counts_to_codes_to_planes(hiplib, oracle, tmp_path):
    import torch
    from tracs_amd import api, synth
    from tracs_amd import device as dev
    L, n = 30011, 6
    alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
    seqs = []
    aln = dev.Alignment(n, L)
    for s in range(n):
        counts = synth.allele_counts(L, seed=100 + s, depth=12 + 6 * s, p_two=0.03)
        counts[5 * s:5 * s + 40] = 0                                                       # uncovered stretch
        keep = bool(s & 1)
        band = (2.0, 4.0) if s == 3 else None
        thr = 0.2 + 0.05 * (s % 3)
        ref = _reference_sequence(oracle, counts, alphas, keep, thr, 3, band)
        codes = dev.posterior_codes_device(torch.from_numpy(counts.view(np.int16)).cuda(), alphas, keep, thr, min_cov=3,
                                           cov_band=band)
        got = dev.codes_to_iupac_device(codes, L).cpu().numpy()
        assert np.array_equal(got, ref), (s, np.where(got != ref)[0][:5])
        assert (ref == ord("N")).any() and (ref == ord("A")).any()
        aln.pack_codes(codes, s)
        seqs.append(ref)
    seqs = np.array(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    fa = os.path.join(str(tmp_path), "codes.fa")
    synth.write_fasta(fa, seqs)
    r, c, ed, names, _, enn = api.pairsnp_arrays([fa], 1, 2147483647, False)
    ri, ci = r.astype(np.int64), c.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    orr, occ, od, onn = oracle.pairsnp_arrays(seqs)
    assert np.array_equal(ed, od) and np.array_equal(enn, onn)


def _counts_for(case, L=40011):
    from tracs_amd import synth
    rng = np.random.default_rng(case["seed"])
    if case.get("flat"):                                                # one allele per site, no sequencing errors
        c = np.zeros((L, 4))
        c[np.arange(L), rng.integers(0, 4, L)] = rng.poisson(case["depth"], L)
    else:
        c = synth.allele_counts(L, seed=300 + case["seed"], depth=case["depth"], p_two=case.get("p_two", 0.02)).astype(np.float64)
    c[rng.random(L) < case.get("p_zero", 0.02)] = 0                     # uncovered sites
    dup = rng.random(L) < case.get("p_dup", 0.0)                        # duplicated regions: ~half coverage outliers
    c[dup] = np.floor(c[dup] / 2.5)
    for i in case.get("poly", ()):
        c[i] = [5, 3, 0, 0]
    return c * case.get("scale", 1)


_CASES = [
    dict(seed=1, depth=14, opts=dict()),                                           # low depth: no outlier band
    dict(seed=2, depth=120, p_dup=0.05, opts=dict()),                              # median > 50: outlier band in play
    dict(seed=3, depth=120, p_dup=0.05, opts=dict(keep_cov_outliers=True)),
    dict(seed=4, depth=60, opts=dict(keep_all=True, min_cov=9, error_threshold=0.03)),
    dict(seed=5, depth=25, opts=dict(consensus=True, min_cov=7)),
    dict(seed=6, depth=30, p_zero=0.8, opts=dict()),                               # < 25 % of the genome covered: skipped
    dict(seed=7, depth=30, p_zero=0.8, opts=dict(consensus=True)),                 # > 75 % N: skipped
    dict(seed=8, depth=8, flat=True, poly=(5, 600, 7000), opts=dict(min_cov=2)),   # <= 5 polymorphic sites: alphas (0,0,0,1)
    # depth beyond uint16 (deep amplicon / viral data; the reference works on float64 counts of any depth): uint32 kernels
    dict(seed=9, depth=120, p_dup=0.05, scale=700, opts=dict()),                   # per-allele counts ~84 000, totals inside the histogram
    dict(seed=10, depth=120, p_dup=0.05, scale=3000, opts=dict()),                 # totals ~360 000: beyond the histogram, sorted order statistics
    dict(seed=11, depth=25, scale=4000, opts=dict(consensus=True, min_cov=7)),
]


@pytest.mark.parametrize("case", _CASES, ids=lambda c: "seed%d" % c["seed"])
def test_call_sequence_matches_restatement(case, hiplib, oracle):
    """counts -> coverage profile -> alphas -> thresholds -> posterior -> coverage rules -> letters, every decision of
    tracs/align.py:476-630, against oracle.call_sequence (numpy median/quantile, scipy digamma, the C posterior filter)."""
    from tracs_amd import align_post
    counts = _counts_for(case)
    got = align_post.call_sequence(counts, **case["opts"])
    fit = oracle.call_sequence(counts, **case["opts"])                              # the oracle's own alpha fit
    want = oracle.call_sequence(counts, alphas=got["alphas"], **case["opts"]) if got["alphas"] is not None else fit
    rs = counts.sum(1)
    assert got["frac_covered"] == np.sum(rs > 0) / len(rs)
    assert got["frac_min_cov"] == np.sum(rs >= case["opts"].get("min_cov", 5)) / len(rs)
    assert got["median_cov"] == np.median(rs[rs > 0])
    if want["sequence"] is None:
        assert got["sequence"] is None and fit["sequence"] is None
    else:
        assert got["sequence"] == want["sequence"].encode()
        assert got["codes"] is not None
    if fit["alphas"] is not None:
        assert np.allclose(got["alphas"], fit["alphas"], rtol=1e-7, atol=1e-12)
        assert got["threshold"] == want["threshold"]
        assert (got["band"] is None) == (want["band"] is None)
        if want["band"] is not None:
            assert got["band"] == tuple(want["band"])
        assert np.array_equal(got["posterior"], want["csv"])                       # what goes into the .csv.gz
    if case["seed"] == 2:
        assert want["band"] is not None and want["band"][1] > want["band"][0]      # the band really masked sites
        inside = (rs <= want["band"][1]) & (rs >= want["band"][0])
        assert inside.sum() > 100
    if case["seed"] == 8:
        assert got["alphas"].tolist() == [0.0, 0.0, 0.0, 1.0]


def test_call_sequence_refuses_counts_it_cannot_narrow(hiplib):
    from tracs_amd import align_post
    from tracs_amd._lib import TracsError
    c = np.full((100, 4), 3.0)
    for bad in (2.0 ** 30, 2.5, -1.0, np.nan):
        x = c.copy()
        x[17, 2] = bad
        with pytest.raises(TracsError):
            align_post.call_sequence(x)


def test_align_post_files(hiplib, oracle, tmp_path):
    """pileup text + reference FASTA -> .csv.gz + .fasta, then `combine` -> the alignment `distance` reads."""
    import gzip
    import subprocess
    import sys
    from tracs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    contigs = [("chr", 30000), ("plasmid", 2500)]
    L = sum(c[1] for c in contigs)
    refseq = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(1).integers(0, 4, L)]
    ref = tmp_path / "ref.fa"
    ref.write_bytes(b">chr the chromosome\n" + refseq[:30000].tobytes() + b"\n>plasmid\n" + refseq[30000:].tobytes() + b"\n")
    dirs = []
    for s in range(2):
        counts = synth.allele_counts(L, seed=900 + s, depth=40).astype(np.int64)
        counts[100 * s:100 * s + 300] = 0
        lines = []
        for i in range(L):
            if counts[i].sum() == 0:
                continue                                                            # uncovered positions have no pileup line
            name, pos = ("chr", i + 1) if i < 30000 else ("plasmid", i - 30000 + 1)
            nz = [k for k in range(4) if counts[i, k]]
            fwd = [int(counts[i, k]) // 2 for k in nz]
            rev = [int(counts[i, k]) - f for k, f in zip(nz, fwd)]
            lines.append("%s\t%d\t%s\t%s\t%d:%s:%s" % (name, pos, chr(refseq[i]), ",".join("ACGT"[k] for k in nz), counts[i].sum(),
                                                      ",".join(map(str, fwd)), ",".join(map(str, rev))))
        d = tmp_path / ("samp%d" % s)
        d.mkdir()
        dirs.append(str(d))
        with gzip.open(d / "pile.txt.gz", "wt") as f:
            f.write("\n".join(lines) + "\n")
        rc = subprocess.run([sys.executable, "-m", "tracs_amd", "align-post", "--pileup", str(d / "pile.txt.gz"), "--reference",
                             str(ref), "--ref-id", "REF1", "-o", str(d), "-p", "samp%d" % s], capture_output=True, text=True, cwd=root)
        assert rc.returncode == 0, rc.stderr
        parsed = oracle.pileup_counts(lines, contigs, True)                          # the CLI default requires both strands
        got = np.loadtxt(gzip.open(d / ("samp%d_posterior_counts_ref_REF1.csv.gz" % s)), delimiter=",")
        fa = (d / ("samp%d_posterior_counts_ref_REF1.fasta" % s)).read_text().split("\n")
        assert fa[0] == ">samp%d_REF1" % s and len(fa[1]) == L and fa[2] == ""
        from tracs_amd import align_post
        res = align_post.call_sequence(parsed)
        want = oracle.call_sequence(parsed, alphas=res["alphas"])
        assert fa[1] == want["sequence"]
        assert got.shape == (L, 4) and np.allclose(got, want["csv"], atol=5.1e-6, rtol=0)
    out = tmp_path / "comb"
    rc = subprocess.run([sys.executable, "-m", "tracs_amd", "combine", "-i"] + dirs + ["-o", str(out)], capture_output=True,
                        text=True, cwd=root)
    assert rc.returncode == 0, rc.stderr
    from tracs_amd import api
    r, c, d_, names, _, nn = api.pairsnp_arrays([str(out / "REF1_combined.fasta.gz")], 1, 2147483647, False)
    assert names == ["samp0", "samp1"] and len(d_) == 1 and nn[0] <= L
