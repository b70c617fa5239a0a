"""BASELINE.json configs[0] and configs[2] as -m gpu tests (the two configurations no GPU test exercised).

configs[0]  `tracs distance` on 10 simulated 100 kb FASTA isolates: the command line end to end, every CSV field against the
            oracle (SNP distance, compared sites, date difference) and the _ref-pinned transcluster rules (tests/ek_parity.py).
configs[2]  10 000 samples x 5 Mbp at FULL size, generated on the device: three disjoint 96-sample blocks (first, middle,
            last rows -- and every cross pair between them) bit-exact against the oracle, size-independent properties over
            all 49 995 000 pairs; once as the consensus alignment the metric is quoted on, once with SURVEY 8d's sprinkling
            of partial IUPAC codes (general encoding: one-hot matrix-core kernel + sparse correction), and the
            counts -> posterior -> codes -> planes front end chained in front of the pair kernel on a sample batch."""
import csv
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev(hiplib):
    import torch
    assert torch.cuda.is_available()
    from tracs_amd import device
    return device


def test_config1_distance_cli_10x100kb(hiplib, oracle, tmp_path):
    from ek_parity import check_trans_dist
    from tracs_amd import synth
    n, L = 10, 100000
    seqs = synth.alignment(n, L, seed=20241022, mu_lineage=1e-4, mu_sample=1e-4, n_lineages=2, p_n=0.01)
    names = ["isolate_%02d" % i for i in range(n)]
    fa = tmp_path / "sim_combined.fasta"
    synth.write_fasta(str(fa), seqs, names=names, width=80)
    iso, days = synth.dates(n, seed=20241022, span_days=120)
    meta = tmp_path / "dates.csv"
    meta.write_text("sample,date\n" + "".join("%s,%s\n" % (nm, d) for nm, d in zip(names, iso)))
    out = tmp_path / "dist.csv"
    rc = subprocess.run([sys.executable, "-m", "tracs_amd", "distance", "--msa", str(fa), "--meta", str(meta), "-o", str(out)],
                        capture_output=True, text=True, cwd=ROOT)
    assert rc.returncode == 0, rc.stderr
    rows = list(csv.reader(open(out)))
    assert rows[0] == ["sampleA", "sampleB", "date difference", "SNP distance", "transmission distance", "expected K",
                       "filtered SNP distance", "sites considered", "MSA file"]
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs)
    assert len(rows) - 1 == n * (n - 1) // 2 == len(ed)
    delta = np.abs(days[er.astype(np.int64)] - days[ec.astype(np.int64)]).astype(np.float64) * 86400.0 / 31556952.0
    p, ek = [], []
    for t, r in enumerate(rows[1:]):
        assert r[0] == names[int(er[t])] and r[1] == names[int(ec[t])]              # row-major pair order (pairsnp.hpp:451-455)
        assert r[2] == str(np.float64(delta[t]))                                    # str(numpy.float64), tracs/distance.py:214
        assert r[3] == str(int(ed[t])) and r[7] == str(int(enn[t]))
        assert r[6] == "NA" and r[8] == "sim"                                       # --filter off with metadata (:204); ref name (:208-209)
        p.append(float(r[4])); ek.append(float(r[5]))
    assert ed.max() > 5 and len(set(delta.tolist())) > 5
    # P(direct) = exp(p0) and E(K) by the transcluster parity rule (p0 1e-6 relative; E(K) per class, tests/ek_parity.py)
    check_trans_dist(oracle, ed.astype(np.int32), delta, 1e-3 * 29903, 73.0, 0.01, np.log(np.array(p)), np.array(ek))


def _generate_with_blocks(dev, synth, n, L, seed, blocks, mutate=None, **kw):
    """Pack the synthetic alignment on the device; keep the ASCII of the sample ranges in `blocks` on the host.
    mutate(rows, first): edits a batch of samples (device uint8 [k, L]) before it is packed."""
    import torch
    aln = dev.Alignment(n, L)
    kept = {}

    def emit(rows, first):
        if mutate is not None:
            mutate(rows, first)
        aln.pack(rows, first=first)
        for b0, b1 in blocks:
            lo, hi = max(b0, first), min(b1, first + rows.shape[0])
            if lo < hi:
                kept[lo] = rows[lo - first:hi - first].cpu().numpy()
    synth.generate_device(n, L, seed, emit, **kw)
    torch.cuda.synchronize()
    idx = np.concatenate([np.arange(b0, b1) for b0, b1 in blocks])
    host = np.concatenate([kept[k] for k in sorted(kept)], axis=0)
    assert host.shape[0] == len(idx)
    return aln, idx, host


@pytest.mark.parametrize("p_partial", [0.0, 0.005], ids=["consensus", "partial-codes"])
def test_config3_full_size(dev, oracle, p_partial):
    import torch
    from tracs_amd import synth
    n, L = 10000, 5000000
    blocks = [(0, 96), (4960, 5056), (n - 96, n)]
    aln, idx, host = _generate_with_blocks(dev, synth, n, L, 20241022 + 2, blocks, mu_lineage=0.0, mu_sample=1e-4, n_lineages=1,
                                           p_n=0.01, p_partial=p_partial)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    torch.cuda.synchronize()
    assert aln.encoding == ("consensus" if p_partial == 0 else "general")
    assert aln.kernel == ("mfma" if p_partial == 0 else "mfma-general")            # the kernels the bench line is about
    # (1) the three blocks and all their cross pairs, bit-exact vs the oracle (41 328 pairs at full length)
    er, ec, ed, enn = oracle.pairsnp_arrays(host, n_threads=max(1, os.cpu_count() or 1))
    gi, gj = idx[er.astype(np.int64)], idx[ec.astype(np.int64)]
    sub = torch.from_numpy(idx).cuda()
    dsub = d[sub][:, sub].cpu().numpy()
    nsub = nn[sub][:, sub].cpu().numpy()
    li, lj = er.astype(np.int64), ec.astype(np.int64)
    assert np.array_equal(dsub[li, lj], ed.astype(np.int32)), "SNP distances differ from the oracle"
    assert np.array_equal(nsub[li, lj], enn.astype(np.int32)), "compared-site counts differ from the oracle"
    assert len(ed) == len(idx) * (len(idx) - 1) // 2 and gi.max() < n and gj.max() == n - 1
    # (2) size-independent properties over all 49 995 000 pairs
    iu = torch.triu_indices(n, n, offset=1, device="cuda")
    dv, nv = d[iu[0], iu[1]], nn[iu[0], iu[1]]
    assert bool((dv >= 0).all()) and bool((dv <= nv).all()) and bool((nv <= L).all())
    mean_d = float(dv.double().mean().item())
    lo, hi = (900.0, 1100.0) if p_partial == 0 else (12000.0, 30000.0)             # E[d] ~ 2 mu L; a random partial code misses the other base at ~40 % of its sites
    assert lo < mean_d < hi, mean_d
    del iu, dv, nv
    # a rectangular block and a row panel computed on their own equal the same cells of the full pass
    d2 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d2, None, row_begin=0, row_end=1000, col_begin=7000)
    assert bool(torch.equal(d2[:1000, 7000:], d[:1000, 7000:])) and int(d2[:, :7000].abs().sum()) == 0
    d2.zero_()
    dev.pairsnp_dense(aln, d2, None, row_begin=6100, row_end=7333)
    assert bool(torch.equal(torch.triu(d2[6100:7333], diagonal=6101), torch.triu(d[6100:7333], diagonal=6101)))
    # (3) thresholded pass (two-pass early-out path): every pair within the threshold identical, the others flagged
    thr = int(mean_d) - 20
    d3 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    n3 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d3, n3, dist_threshold=thr)
    keep = torch.triu(d <= thr, diagonal=1)
    assert int(keep.sum()) > 1000
    assert bool(torch.equal(d3[keep], d[keep])) and bool(torch.equal(n3[keep], nn[keep]))
    rest = torch.triu(d > thr, diagonal=1)
    assert bool(((d3[rest] < 0) | (d3[rest] > thr)).all())
    aln.close()


@pytest.mark.parametrize("p_partial", [0.0, 0.005, -1.0], ids=["consensus", "partial-codes", "coverage"])
def test_config3_full_size_filter(dev, oracle, p_partial):
    """`tracs distance --filter` at the metric's size (src/pairsnp.hpp:251-318 on every emitted pair, :405-413): three 32-sample
    blocks (first rows, middle, last) and all their cross pairs -- 4 560 pairs at full length -- against the oracle's filter_recomb,
    with planted runs of substitutions (what the filter exists to remove) in some of them.  Consensus alignment: the departure
    lists, ALL 49 995 000 pairs filtered in one call; with partial codes (and on bench.py's `coverage` workload: what `tracs align`
    writes) the lists outgrow a wave's LDS and the pairs take the tiled merge over global memory (the block's pairs only)."""
    import torch
    import bench
    from tracs_amd import synth
    n, L = 10000, 5000000
    blocks = [(0, 32), (4984, 5016), (n - 32, n)]
    planted = {3: (1000, 400), 17: (2500000, 250), 4990: (4999700, 300), 5001: (0, 350), n - 2: (777777, 500), n - 30: (2500100, 200)}
    lut = torch.arange(256, dtype=torch.uint8, device="cuda")
    for a, b in zip(b"ACGT", b"CGTA"):
        lut[a] = b

    def mutate(rows, first):
        for s, (at, w) in planted.items():
            if first <= s < first + rows.shape[0]:
                seg = rows[s - first, at:at + w]
                rows[s - first, at:at + w] = lut[seg.long()]

    kw = bench.synth_kw(0.0, "coverage") if p_partial < 0 else dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=p_partial)
    aln, idx, host = _generate_with_blocks(dev, synth, n, L, 20241022 + 2, blocks, mutate=mutate, **kw)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    er, ec, ed, enn = oracle.pairsnp_arrays(host, n_threads=max(1, os.cpu_count() or 1))
    ef = oracle.filter_recomb_pairs(host, er, ec, max(1, os.cpu_count() or 1)).astype(np.int32)
    gi = torch.from_numpy(idx[er.astype(np.int64)].astype(np.int32)).cuda()
    gj = torch.from_numpy(idx[ec.astype(np.int64)].astype(np.int32)).cuda()
    gd = d[gi.long(), gj.long()].contiguous()
    assert np.array_equal(gd.cpu().numpy(), ed.astype(np.int32))
    # the block's pairs on their own (any order of pairs is a valid call)
    got = dev.filter_recomb_pairs(aln, gi, gj, gd).cpu().numpy()
    assert np.array_equal(got, ef), np.where(got != ef)[0][:10]
    hit = np.isin(idx[er.astype(np.int64)], list(planted)) | np.isin(idx[ec.astype(np.int64)], list(planted))
    assert (ef <= ed).all()
    if p_partial >= 0:                                          # (coverage: a planted run may sit in one of the sample's gaps)
        assert (ed[hit].astype(np.int64) - ef[hit] >= 150).all()                       # the planted runs are filtered out
    info = dev.filter_index_info(aln)
    assert info["lists"] and info["entries"] > 4_000_000
    if p_partial == 0:
        assert info["longest_list"] <= 1024                     # every pair from the lists (the merge-path kernel)
        rows, cols, dd, _ = dev.coo_from_dense(d, nn, n)
        assert rows.numel() == n * (n - 1) // 2
        filt = dev.filter_recomb_pairs(aln, rows, cols, dd)
        assert bool((filt <= dd).all()) and bool((filt >= 0).all())
        fm = torch.zeros((n, n), dtype=torch.int32, device="cuda")
        fm[rows.long(), cols.long()] = filt
        assert np.array_equal(fm[gi.long(), gj.long()].cpu().numpy(), ef)
        # the filtered distance only depends on the pair: a thresholded emission gives the same values
        thr = int(dd.double().mean().item()) - 20
        r2, c2, d2, _ = dev.coo_from_dense(d, nn, n, thr)
        assert 1000 < r2.numel() < rows.numel()
        assert bool(torch.equal(dev.filter_recomb_pairs(aln, r2, c2, d2), fm[r2.long(), c2.long()]))
    else:
        assert info["longest_list"] > 4096                      # the lists do not fit a wave's LDS: these pairs were scanned
    aln.close()


# what each workload of bench.py's `sensitivity` must exercise at 10 000 x 5 Mbp (the cost model's decisions at the metric's size)
FULL_SIZE_PATHS = {
    "lineage": lambda c, src, ls: src[2] > 4_000_000 and ls["row_splits"] > 1,                   # N in 1 / 21 of the samples: long rows, cut over workgroups
    "divergent": lambda c, src, ls: c[2] > 4_000_000 and src[2] > 4_000_000 and ls["p_entries"] > 40_000_000,   # ~10 listed samples at every site
    "clean": lambda c, src, ls: src[2] == 0 and src[0] == 0 and c[3] > 4_000_000 and not ls.get("bitmaps", False),   # no N: neither lists nor counting pass for nn
    "gappy": lambda c, src, ls: src[1] is True and src[2] == 0 and c[2] > 2_000_000,              # every site on the matrix cores, in place; minority lists of ~1000 N samples
    "runs": lambda c, src, ls: src[0] > 50_000 and src[1] is False and src[2] > 10_000,           # counted (re-packed) and listed sites side by side
    # what `tracs align` writes (30 % N per sample in runs of its own, partial codes, two lineages): every site minority by k^2, no N
    # lists, the N x listed terms from the two one-plane passes (nw_gram), compared sites from the N plane in place
    "coverage": lambda c, src, ls: c[2] > 4_900_000 and c[0] < 20_000 and src[1] is True and src[2] == 0 and ls["p_entries"] > 100_000_000,
}


@pytest.mark.parametrize("workload", sorted(FULL_SIZE_PATHS))
def test_config3_full_size_other_workloads(dev, oracle, workload):
    """The other workloads of bench.py's `sensitivity` at the metric's size, against the ORACLE (the bench compares them with the
    library's own dense pass only): three 64-sample blocks -- the first rows, the rows around the middle, the last -- and all their
    cross pairs, 18 336 pairs at full length (src/pairsnp.hpp:395-420), and which paths the cost model took there."""
    import torch
    import bench
    from tracs_amd import synth
    n, L = 10000, 5000000
    blocks = [(0, 64), (4968, 5032), (n - 64, n)]
    aln, idx, host = _generate_with_blocks(dev, synth, n, L, 20241022 + 2, blocks, **bench.synth_kw(0.0, workload))
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    torch.cuda.synchronize()
    classes, src, ls = aln.site_classes, aln.count_source, aln.list_stats
    assert classes is not None and aln.kernel == ("mfma-general" if workload == "coverage" else "mfma"), (classes, aln.kernel)
    assert FULL_SIZE_PATHS[workload](classes, src, ls), (workload, classes, src, ls)
    assert aln.nw_gram == (workload == "coverage") and aln.nw_form == ("ns-rows" if workload == "coverage" else None)
    er, ec, ed, enn = oracle.pairsnp_arrays(host, n_threads=max(1, os.cpu_count() or 1))
    sub = torch.from_numpy(idx).cuda()
    dsub, nsub = d[sub][:, sub].cpu().numpy(), nn[sub][:, sub].cpu().numpy()
    li, lj = er.astype(np.int64), ec.astype(np.int64)
    assert len(ed) == len(idx) * (len(idx) - 1) // 2
    assert np.array_equal(dsub[li, lj], ed.astype(np.int32)), "SNP distances differ from the oracle"
    assert np.array_equal(nsub[li, lj], enn.astype(np.int32)), "compared-site counts differ from the oracle"
    # the same cells from row panels computed on their own (a multi-GPU rank's call): one around the middle, the last rows
    for r0, r1 in ((4900, 5100), (n - 300, n)):
        dp = torch.zeros((r1 - r0, n), dtype=torch.int32, device="cuda")
        npn = torch.zeros_like(dp)
        dev.pairsnp_dense(aln, dp, npn, row_begin=r0, row_end=r1, base_row=r0)
        up = torch.triu(torch.ones((r1 - r0, n), dtype=torch.bool, device="cuda"), diagonal=r0 + 1)
        assert bool(torch.equal(dp[up], d[r0:r1][up])) and bool(torch.equal(npn[up], nn[r0:r1][up])), (r0, r1)
    aln.close()


def test_config3_dm_frontend_chain(dev, oracle):
    """counts -> posterior filter -> 4-bit codes -> planes -> pairsnp on a batch of full-length samples, against the same
    chain through the oracle (calculate_posteriors -> IUPAC letters -> pack -> pair loop)."""
    import torch
    from tracs_amd import synth
    batch, L = 6, 5000000
    alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
    lut = np.frombuffer(b"XACMGRSVTWYHKDBN", dtype=np.uint8)
    aln = dev.Alignment(batch, L)
    stride = ((L + 1) // 2 + 15) // 16 * 16
    codes = torch.zeros((batch, stride), dtype=torch.uint8, device="cuda")
    letters = np.empty((batch, L), dtype=np.uint8)
    base = synth.allele_counts(L, seed=77, depth=30, p_two=0.01)
    rng = np.random.default_rng(78)
    for b in range(batch):
        c = base.copy()
        m = rng.random(L) < 0.002                                   # each sample differs from the shared profile at a few sites
        c[m] = c[m][:, rng.permutation(4)]
        c[rng.random(L) < 0.01] = 0
        post = oracle.calculate_posteriors(c.astype(np.float64), alphas, False, 0.05)
        mask = ((post > 0).astype(np.uint8) * np.array([1, 2, 4, 8], np.uint8)).sum(1)
        letters[b] = lut[mask]                                      # tracs/align.py:616-622 ('X' for an empty mask)
        codes[b, :(L + 1) // 2] = dev.posterior_codes_device(torch.from_numpy(c.view(np.int16)).cuda(), alphas, False, 0.05)
    aln.pack_codes(codes, 0)                                        # one launch for the batch
    d = torch.zeros((batch, batch), dtype=torch.int32, device="cuda")
    nn = torch.zeros((batch, batch), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    er, ec, ed, enn = oracle.pairsnp_arrays(letters, n_threads=8)
    li, lj = er.astype(np.int64), ec.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[li, lj], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[li, lj], enn.astype(np.int32))
    assert ed.max() > 100
