"""GPU path against the committed golden fixtures, the device-resident entry points, and size-independent
properties at larger sizes.  Everything goes through the C ABI (ctypes)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_mod(hiplib):
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def api(torch_mod):
    from tracs_amd import api
    return api


@pytest.fixture(scope="module")
def dev(torch_mod):
    from tracs_amd import device
    return device


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


def test_trans_dist_golden(api, oracle, golden_dir):
    """p0 and E(K) against the goldens produced by oracle/_ref (the reference's transcluster.hpp compiled with setup.py's flags).
    E(K) by stopping-rule class (tests/ek_parity.py): 'well' 1e-6 vs the reference; 'ill' (the reference's stop is decided by
    rounding noise; two builds of the reference itself disagree there) bounded at 5 % vs the reference and bracketed by the
    series vs the oracle; 'saturated' (k = 10000: the reference reads past its 10 000-entry lgamma table, src/transcluster.hpp
    :140 with :253-258, so its value depends on heap contents) vs the oracle only.  The per-class deviations are written to
    gpurun_out/ek_golden_classes.json (committed as profiles/r02/ek_golden_classes.json)."""
    from ek_parity import check_ek
    g = _load(golden_dir, "transcluster_golden.json")
    counts, dev = {}, {"well": [], "ill": [], "saturated": []}
    ref_inf = 0
    for grid in g["trans_dist"]:
        N, delta = np.array(grid["N"], np.int32), np.array(grid["delta"])
        p0, ek = api.trans_dist_arrays(N, delta, grid["lamb"], grid["beta"], grid["thr"])
        assert np.allclose(p0, grid["p0"], rtol=1e-6, atol=0)                  # vs the reference build (log-likelihood)
        assert np.max(np.abs(p0 - grid["p0"]) / np.abs(grid["p0"])) < 1e-9
        for i, cls in enumerate(grid["conditioning"]):
            ref = grid["eK"][i]
            if not np.isfinite(ref):                                            # the reference's exp() overflowed; ours stays finite
                assert cls == "ill" and np.isfinite(ek[i])
                ref_inf += 1
            else:
                rel = abs(ek[i] - ref) / max(abs(ref), 1e-300)
                dev[cls].append(rel)
                if cls == "well":
                    assert rel <= 1e-6, (N[i], delta[i], ek[i], ref)
                elif cls == "ill":
                    assert rel <= 0.05, (N[i], delta[i], ek[i], ref)
            check_ek(oracle, N[i], delta[i], grid["lamb"], grid["beta"], grid["thr"], ek[i], counts)   # vs the oracle
    assert counts["well"] > 250 and counts.get("ill", 0) >= 10
    # the 'ill' keys of this grid: every one within 1e-6 of the shipped reference but ONE (4 % off: a key whose stop the reference
    # decides by rounding noise -- where its two builds agree the outbreak grid below asserts 1e-3)
    ill_sorted = sorted(dev["ill"])
    assert len(ill_sorted) >= 10 and ill_sorted[-2] <= 1e-6 and ill_sorted[-1] <= 0.05, ill_sorted[-3:]
    summary = {c: {"keys": len(v), "max_rel_vs_ref": float(np.max(v)) if v else None,
                   "n_above_1e-6": int(np.sum(np.array(v) > 1e-6))} for c, v in dev.items()}
    summary["ill"]["reference_returned_inf"] = ref_inf
    summary["saturated"]["note"] = "reference value depends on heap contents (out-of-bounds lgamma read); not a parity target"
    os.makedirs(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out", "ek_golden_classes.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


def test_trans_dist_outbreak_grid(api, golden_dir):
    """4 000 outbreak-scale keys at the CLI defaults (N <= 80 SNPs, 1..730 days, lamb = 29.903, beta = 73, precision 0.01:
    tests/golden/make_golden.py outbreak) against the reference as shipped (oracle/_ref, -ffast-math) AND the same source compiled
    IEEE-strict.  86 % of the grid is 'well' conditioned: E(K) at 1e-6 (observed 1e-12).  14 % is 'ill' -- few SNPs over a gap of
    200 days or more: the bound upper ~ e^(lamb delta) is so large that the stop is decided by the last bits of exp(elprob) --
    and there the reference is not one function: the shipped build returns inf / nan on 28 % of those keys, the strict build on
    more, and where both are finite they disagree with each other on one key in six (by up to 79 %).  What can be asserted: where
    the two builds ARE finite and agree to 1e-6 we are within 1e-3 of them (observed <= 1.1e-4, 97 % within 1e-6), always finite,
    and p0 -- which has no stopping rule -- at 1e-9 everywhere.  The counts go to gpurun_out/ek_outbreak.json."""
    g = _load(golden_dir, "transcluster_outbreak_golden.json")
    N = np.array(g["N"], np.int32)
    delta = np.array(g["days"], np.float64) * 86400.0 / 31556952.0
    p0, ek = api.trans_dist_arrays(N, delta, g["lamb"], g["beta"], g["thr"])
    assert np.max(np.abs(p0 - np.array(g["p0"])) / np.abs(np.array(g["p0"]))) < 1e-9 and np.isfinite(ek).all()
    ref = np.array([np.nan if v is None else v for v in g["eK"]])
    strict = np.array([np.nan if v is None else v for v in g["eK_strict_build"]])
    cls = np.array(g["conditioning"])
    well = cls == "well"
    assert well.sum() > 3000 and np.max(np.abs(ek[well] - ref[well]) / np.abs(ref[well])) <= 1e-6
    ill = (cls == "ill") & np.isfinite(ref) & np.isfinite(strict)
    agree = ill & (np.abs(strict - ref) <= 1e-6 * np.abs(ref))
    assert agree.sum() >= 200                                                   # the table of INTEGRATION.md 4 rests on these
    rel = np.abs(ek[agree] - ref[agree]) / np.abs(ref[agree])
    assert rel.max() <= 1e-3 and (rel <= 1e-6).mean() >= 0.95, (rel.max(), (rel <= 1e-6).mean())
    summary = {"keys": int(len(N)), "well": int(well.sum()), "ill": int((cls == "ill").sum()),
               "ill_reference_not_finite": int(((cls == "ill") & ~np.isfinite(ref)).sum()),
               "ill_strict_build_not_finite": int(((cls == "ill") & ~np.isfinite(strict)).sum()),
               "ill_both_builds_finite": int(ill.sum()), "ill_builds_agree_1e-6": int(agree.sum()),
               "of_those_ours_within_1e-6": int((rel <= 1e-6).sum()), "of_those_ours_max_rel": float(rel.max()),
               "ill_finite_reference_ours_within_1e-6": int((np.abs(ek - ref)[(cls == "ill") & np.isfinite(ref)] <= 1e-6 * np.abs(ref[(cls == "ill") & np.isfinite(ref)])).sum()),
               "ill_finite_reference": int(((cls == "ill") & np.isfinite(ref)).sum()),
               "max_rel_between_the_builds_where_both_finite": float(np.max(np.abs(strict - ref)[ill] / np.abs(ref[ill])))}
    os.makedirs(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out", "ek_outbreak.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


def test_trans_dist_golden_large_n(api, golden_dir):
    """Keys with 128 .. 1 600 SNPs -- the ones the wave-per-key kernel takes from their first term (csrc/transcluster.hip) --
    against oracle/_ref goldens (tests/golden/make_golden.py large-n): all 'well' conditioned, p0 and E(K) at 1e-6 relative
    (BASELINE.json north_star), observed far tighter."""
    g = _load(golden_dir, "transcluster_golden_large_n.json")
    worst_p, worst_e = 0.0, 0.0
    for grid in g["trans_dist"]:
        N, delta = np.array(grid["N"], np.int32), np.array(grid["delta"])
        assert set(grid["conditioning"]) == {"well"}
        p0, ek = api.trans_dist_arrays(N, delta, grid["lamb"], grid["beta"], grid["thr"])
        rp = np.abs(p0 - grid["p0"]) / np.abs(grid["p0"])
        re = np.abs(ek - grid["eK"]) / np.abs(grid["eK"])
        worst_p, worst_e = max(worst_p, float(rp.max())), max(worst_e, float(re.max()))
        assert rp.max() < 1e-9 and re.max() < 1e-6, (float(rp.max()), float(re.max()))
    print("large-N keys: max relative deviation from the reference build: p0 %.2e, E(K) %.2e" % (worst_p, worst_e))
    assert worst_e < 1e-8


def test_lprob_golden(api, golden_dir):
    from scipy.special import gammaln
    g = _load(golden_dir, "transcluster_golden.json")["lprob"]
    lg = gammaln(np.arange(g["lgamma_len"]))
    for r in g["rows"]:
        got = api.lprob_k_given_N(r["N"], r["k"], r["delta"], r["lamb"], r["beta"], lg)
        assert np.allclose(got, r["lprob_k_given_N"], rtol=1e-6, atol=1e-9)
        assert np.allclose(got, r["lprob_k_given_N"], rtol=1e-10, atol=1e-11)


def test_posteriors_golden(api, golden_dir):
    z = np.load(os.path.join(golden_dir, "posteriors_golden.npz"))
    counts = z["counts"]
    for i, m in enumerate(json.loads(str(z["meta"]))):
        got = api.calculate_posteriors(counts, m["alphas"], m["keep"], m["threshold"])
        exp = z["post_%d" % i]
        assert np.max(np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)) <= 4e-16, m


def test_fasta_reader_golden(api, oracle, golden_dir, tmp_path):
    g = _load(golden_dir, "kseq_golden.json")
    for name, case in g.items():
        if case.get("crash"):
            continue
        p = os.path.join(str(tmp_path), name)
        with open(p, "wb") as fh:
            fh.write(case["text"].encode("latin-1"))
        recs = case["records"]
        if case["rc"] == -2:
            with pytest.raises(RuntimeError, match="Error reading FASTA!"):
                api.pairsnp_arrays([p], 1, 2147483647, False)
            continue
        if len({len(r[1]) for r in recs}) > 1:
            with pytest.raises(RuntimeError, match="variable sequence lengths"):
                api.pairsnp_arrays([p], 1, 2147483647, False)
            continue
        r, c, d, names, _, nn = api.pairsnp_arrays([p], 1, 2147483647, False)
        assert names == [x[0] for x in recs], name
        if len(recs) >= 2:
            seqs = np.array([np.frombuffer(x[1].encode("latin-1"), np.uint8) for x in recs])
            er, ec, ed, enn = oracle.pairsnp_arrays(seqs)
            assert np.array_equal(d, ed) and np.array_equal(nn, enn), name


def test_pairsnp_fixture(api, golden_dir, tmp_path):
    from tracs_amd import synth
    g = _load(golden_dir, "pairsnp_unpinned.json")
    for name, c in g["cases"].items():
        seqs = np.array([np.frombuffer(s.encode("ascii"), np.uint8) for s in c["seqs"]])
        td = str(tmp_path)
        if c["n0"] is None:
            fa = os.path.join(td, "a.fa")
            synth.write_fasta(fa, seqs)
            files = [fa]
        else:
            fa, fb = os.path.join(td, "a.fa"), os.path.join(td, "b.fa")
            synth.write_fasta(fa, seqs[:c["n0"]])
            synth.write_fasta(fb, seqs[c["n0"]:])
            files = [fa, fb]
        out = api.pairsnp(fasta=files, n_threads=1, dist=c["dist"], filter=False)
        assert out[0] == c["rows"] and out[1] == c["cols"] and out[2] == c["d"] and out[5] == c["nn"], name


@pytest.mark.parametrize("path", ["device", "arrays"])
def test_cli_end_to_end_vs_reference_driver(api, golden_dir, tmp_path, monkeypatch, path):
    """`tracs distance` + `tracs cluster` on the GPU against the CSVs the reference's own drivers wrote -- through the device-resident
    path of the single-GPU command (tracs_distance_open / _run: results on the device until the CSV rows, the --filter runs included:
    the filter and the transmission model of the filtered distances per emitted pair on the device) and through the array path (TRACS.pairsnp-shaped arrays -> calculate_trans_prob -> rows)."""
    if path == "arrays":
        monkeypatch.setenv("TRACS_DISTANCE_ARRAYS", "1")
    else:
        monkeypatch.delenv("TRACS_DISTANCE_ARRAYS", raising=False)
    from test_host_logic import _materialise, _rows_equal
    from tracs_amd import cluster, distance
    pyref = _load(golden_dir, "python_reference_golden.json")
    td = str(tmp_path)
    _materialise(pyref, td)
    for run in ("meta", "nometa", "meta_thr", "msadb", "filter", "filter_nometa"):
        r = pyref["distance"]["runs"][run]
        out = os.path.join(td, run + ".csv")
        monkeypatch.setattr(sys, "argv", ["d"] + [a.replace("TMP", td) for a in r["argv"]] + ["-o", out, "--loglevel", "ERROR"])
        distance.main()
        _rows_equal(open(out).read(), r["csv"])
    for key, exp in pyref["cluster"]["runs"].items():
        col, thr = key.split("_")
        cluster._ids.clear()
        out = os.path.join(td, "c.csv")
        monkeypatch.setattr(sys, "argv", ["c", "-d", os.path.join(td, "meta.csv"), "-o", out, "-c", thr, "-D", col,
                                          "--loglevel", "ERROR"])
        cluster.main()
        assert open(out).read() == exp, key
    cluster._ids.clear()


def test_cli_device_path_in_small_batches_equals_array_path(api, tmp_path):
    """The device-resident path of `tracs distance` (tracs_distance_run) with its device-to-host batches cut to 1 000 rows -- 20 of
    them, two in flight -- against the array path on the same alignment: the same rows in the same order; SNP distance, compared
    sites, names and date difference identical as text, P(direct) and E(K) to 10^-9 (exp on the device / in numpy; the dense route sums long series from per-gap
    tables, the array route per key).  With a SNP
    threshold and -K, with a database file, and without metadata."""
    import subprocess
    from tracs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, L = 203, 20000
    seqs = synth.alignment(n, L, seed=37, mu_lineage=3e-3, mu_sample=4e-4, p_n=0.02, p_partial=0.003)
    names = ["s%03d" % i for i in range(n)]
    fa, db, meta = tmp_path / "aln_combined.fasta", tmp_path / "db.fasta", tmp_path / "dates.csv"
    synth.write_fasta(str(fa), seqs[:150], names=names[:150])
    synth.write_fasta(str(db), seqs[150:], names=names[150:])
    iso, _ = synth.dates(n, seed=37, span_days=300)
    meta.write_text("name,date\n" + "".join("%s,%s\n" % (a, b) for a, b in zip(names, iso)))
    for tag, extra in (("meta", ["--meta", str(meta)]), ("thr", ["--meta", str(meta), "-D", "150", "-K", "300", "--msa-db", str(db)]), ("nometa", []),
                       ("filter", ["--meta", str(meta), "--filter"]), ("filter_thr_nometa", ["--filter", "-D", "150", "--msa-db", str(db)])):
        outs = {}
        for path in ("device", "arrays"):
            out = tmp_path / ("%s_%s.csv" % (tag, path))
            env = dict(os.environ, TRACS_DISTANCE_BATCH_ROWS="1000")
            if path == "arrays":
                env["TRACS_DISTANCE_ARRAYS"] = "1"
            rc = subprocess.run([sys.executable, "-m", "tracs_amd", "distance", "--msa", str(fa), "-o", str(out), "--loglevel", "ERROR"] + extra,
                                capture_output=True, text=True, cwd=root, env=env, timeout=600)
            assert rc.returncode == 0, rc.stdout[-2000:] + rc.stderr[-3000:]
            outs[path] = open(out).read().split("\n")
        a, b = outs["device"], outs["arrays"]
        assert len(a) == len(b) and a[0] == b[0] and len(a) > (10000 if tag == "meta" else 300)
        for x, y in zip(a[1:], b[1:]):
            if x == y:
                continue
            fx, fy = x.split(","), y.split(",")
            assert fx[:4] == fy[:4] and fx[6:] == fy[6:], (x, y)
            for c in (4, 5):
                assert abs(float(fx[c]) - float(fy[c])) <= 1e-9 * abs(float(fy[c])) + 1e-300, (x, y)


def test_drop_in_module_names():
    import TRACS
    for f in ("pairsnp", "trans_dist", "lprob_k_given_N", "calculate_posteriors"):
        assert callable(getattr(TRACS, f))


# ---- device-resident entry points --------------------------------------------------------------------------
def test_dense_coo_and_transcluster_device(dev, oracle, torch_mod):
    from ek_parity import check_ek
    from tracs_amd import synth
    torch = torch_mod
    n, L = 333, 7000
    seqs = synth.alignment(n, L, seed=31, mu_lineage=0.004, mu_sample=0.001, p_n=0.02, p_partial=0.01, p_lower=0.05)
    aln = dev.Alignment(n, L)
    aln.pack(seqs[:100], first=0)                                  # host pointer
    aln.pack(torch.from_numpy(seqs[100:]).cuda(), first=100)       # device pointer
    d = torch.full((n, n), -1, dtype=torch.int32, device="cuda")
    nn = torch.full((n, n), -1, dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    dh, nh = d.cpu().numpy(), nn.cpu().numpy()
    assert np.array_equal(dh[ri, ci], ed.astype(np.int32)) and np.array_equal(nh[ri, ci], enn.astype(np.int32))
    assert (dh[np.tril_indices(n)] == -1).all()                    # nothing outside the cell set is written
    # d only (ncomp = NULL): the 5-op kernel
    d2 = torch.full((n, n), -1, dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d2, None, row_begin=50, row_end=200, col_begin=120)
    sel = (ri >= 50) & (ri < 200) & (ci >= 120)
    d2h = d2.cpu().numpy()
    assert np.array_equal(d2h[ri[sel], ci[sel]], ed[sel].astype(np.int32)) and (d2h != -1).sum() == sel.sum()
    # COO
    for thr in (2147483647, 25, 3, -1):
        rows, cols, dd, nc = dev.coo_from_dense(d, nn, n, dist_threshold=thr)
        xr, xc, xd, xn = oracle.pairsnp_arrays(seqs, dist=thr)
        assert np.array_equal(rows.cpu().numpy(), xr.astype(np.int32)) and np.array_equal(cols.cpu().numpy(), xc.astype(np.int32))
        assert np.array_equal(dd.cpu().numpy(), xd.astype(np.int32)) and np.array_equal(nc.cpu().numpy(), xn.astype(np.int32))
    # transcluster on the dense block, delta from integer days
    _, days = synth.dates(n, seed=31, span_days=200)
    p = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    e = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    dev.trans_dist_dense(d, n, torch.from_numpy(days).cuda(), 5.3, 6.0, 0.01, p, e, exp_p0=True, dist_threshold=60)
    keep = ed <= 60
    delta = np.abs(days[ri] - days[ci]).astype(np.float64) * 86400.0 / 31556952.0
    ep0, _ = oracle.trans_dist(ed[keep].astype(np.int32), delta[keep], 5.3, 6.0, 0.01)
    ph, eh = p.cpu().numpy(), e.cpu().numpy()
    assert np.allclose(ph[ri[keep], ci[keep]], np.exp(ep0), rtol=1e-6, atol=0)
    seen = set()
    for t in np.where(keep)[0][::7]:
        key = (int(ed[t]), float(delta[t]))
        if key not in seen:
            seen.add(key)
            check_ek(oracle, key[0], key[1], 5.3, 6.0, 0.01, eh[ri[t], ci[t]])
    assert (ph[ri[~keep], ci[~keep]] == 0).all()                   # cells above the SNP threshold are skipped
    # two row panels in one pass (the multi-GPU partition's shape) == the full pass on those rows
    p2 = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    e2 = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    dev.trans_dist_dense_ranges(d, n, torch.from_numpy(days).cuda(), 5.3, 6.0, 0.01, p2, e2, [(0, 40), (250, 333)],
                                exp_p0=True, dist_threshold=60)
    rows = list(range(0, 40)) + list(range(250, 333))
    assert torch.equal(p2[rows], p[rows]) and torch.equal(e2[rows], e[rows])
    assert float(p2[40:250].abs().sum()) == 0.0
    aln.close()


def test_encoding_selection_and_consensus_parity(dev, oracle, torch_mod):
    """ACGT + anything-that-means-N (N, n, -, ?, lower case) takes the 3-plane consensus kernel; a single partial
    IUPAC code anywhere sends the whole alignment to the general kernel.  Both must equal the oracle."""
    from tracs_amd import synth
    torch = torch_mod
    n, L = 150, 9000
    seqs = synth.alignment(n, L, seed=41, mu_lineage=0.01, mu_sample=0.004, p_n=0.03, p_lower=0.2, p_other=0.02)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    assert aln.encoding is None
    for with_nn in (True, False):
        d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
        nn = torch.zeros((n, n), dtype=torch.int32, device="cuda") if with_nn else None
        dev.pairsnp_dense(aln, d, nn)
        assert aln.encoding == "consensus"
        assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
        if with_nn:
            assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    seqs2 = seqs.copy()
    seqs2[77, 4321] = ord("r")                                     # one partial code
    aln.pack(seqs2[77:78], first=77)
    assert aln.encoding is None
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    assert aln.encoding == "general"
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs2)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32)) and np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    aln.close()


def test_split_group_range_small_n_long_L(dev, oracle, torch_mod):
    """Few tiles + long alignment: the group range is split over workgroups (atomic accumulation path)."""
    from tracs_amd import synth
    torch = torch_mod
    n, L = 40, 300000
    for p_partial, enc in ((0.0, "consensus"), (0.001, "general")):
        seqs = synth.alignment(n, L, seed=77, mu_lineage=2e-4, mu_sample=1e-4, p_n=0.01, p_partial=p_partial)
        aln = dev.Alignment(n, L)
        aln.pack(seqs)
        d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
        nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
        for _ in range(2):                                        # twice: the init pass must reset the cells
            dev.pairsnp_dense(aln, d, nn)
        assert aln.encoding == enc
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=8)
        ri, ci = er.astype(np.int64), ec.astype(np.int64)
        assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
        assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
        aln.close()


def test_properties_at_scale(dev, oracle, torch_mod):
    """1 500 x 200 kb: properties that need no oracle at full size + an oracle-checked sub-block."""
    from tracs_amd import synth
    torch = torch_mod
    n, L = 1500, 200000
    seqs = synth.alignment(n, L, seed=12, mu_lineage=5e-4, mu_sample=1e-4, p_n=0.01, p_partial=0.002)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    iu = torch.triu_indices(n, n, offset=1, device="cuda")
    dv, nv = d[iu[0], iu[1]], nn[iu[0], iu[1]]
    assert bool((dv <= nv).all()) and bool((nv <= L).all()) and bool((dv >= 0).all())
    # permutation invariance: reversing the sample order transposes the pair set
    aln2 = dev.Alignment(n, L)
    aln2.pack(np.ascontiguousarray(seqs[::-1]))
    d2 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln2, d2, None)
    full = d + d.t()
    full2 = d2 + d2.t()
    assert bool(torch.equal(full, torch.flip(full2, dims=(0, 1))))
    # two-block consistency: rows [0, 700) x cols >= 700 computed as a rectangular block
    d3 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d3, None, row_begin=0, row_end=700, col_begin=700)
    assert bool(torch.equal(d3[:700, 700:], d[:700, 700:])) and int(d3[:, :700].abs().sum()) == 0
    # oracle on a 96-sample sub-block
    sub = np.sort(np.random.default_rng(1).choice(n, 96, replace=False))
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs[sub], n_threads=8)
    gi, gj = sub[er.astype(np.int64)], sub[ec.astype(np.int64)]
    assert np.array_equal(d.cpu().numpy()[gi, gj], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[gi, gj], enn.astype(np.int32))


def test_posterior_codes_and_device_posteriors(dev, oracle, torch_mod):
    from tracs_amd import synth
    torch = torch_mod
    L = 100001
    counts = synth.allele_counts(L, seed=3, depth=18, p_two=0.05)
    counts[:10] = 0
    alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
    for keep in (False, True):
        post = oracle.calculate_posteriors(counts.astype(np.float64), alphas, keep, 0.02)
        got = dev.calculate_posteriors_device(torch.from_numpy(counts.astype(np.float64)).cuda(), alphas, keep, 0.02)
        assert np.array_equal(got.cpu().numpy(), post)
        codes = dev.posterior_codes_device(torch.from_numpy(counts.view(np.int16)).cuda(), alphas, keep, 0.02).cpu().numpy()
        mask = ((post > 0).astype(np.uint8) * np.array([1, 2, 4, 8], np.uint8)).sum(1).astype(np.uint8)
        exp = np.zeros((L + 1) // 2, np.uint8)
        exp |= mask[0::2]
        exp[:L // 2] |= (mask[1::2] << 4)
        assert np.array_equal(codes, exp)


def test_posterior_codes_on_the_threshold(dev, oracle, torch_mod):
    """posterior_codes_kernel screens `post > threshold` in single precision and sends a site to the exact f64 route (the
    reference's own divide and `<=`, src/dmultinomial.hpp:59-82) only when a cell lies within 2^-14 of the threshold.  Thresholds
    that ARE posterior values of the table (so `post <= threshold` holds with equality for every site with that row), thresholds
    one ulp either side, and parameters outside the screen's assumptions (zero alphas from the degenerate fit, zero / negative
    thresholds, uint32 counts beyond 2^24) must all give the oracle's mask."""
    from tracs_amd import synth
    torch = torch_mod
    L = 40001
    counts = synth.allele_counts(L, seed=9, depth=25, p_two=0.1)
    counts[:7] = 0
    cases = [([20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1], None),
             ([1.0, 0.0, 0.0, 0.0], 0.05), ([0.0, 0.0, 0.0, 1.0], 0.3),           # find_dirichlet_priors' degenerate answer, any order
             ([20.8, 4.4, 0.9, 0.1], 0.0), ([20.8, 4.4, 0.9, 0.1], -0.5), ([3.0, 3.0, 3.0, 3.0], 0.25)]
    bits = np.array([1, 2, 4, 8], np.uint8)

    def expect(c, alphas, keep, thr):
        post = oracle.calculate_posteriors(c.astype(np.float64), alphas, keep, thr)
        mask = ((post > 0).astype(np.uint8) * bits).sum(1).astype(np.uint8)
        exp = np.zeros((len(c) + 1) // 2, np.uint8)
        exp |= mask[0::2]
        exp[:len(c) // 2] |= (mask[1::2] << 4)
        return exp
    c16 = torch.from_numpy(counts.view(np.int16)).cuda()
    c32 = torch.from_numpy(counts.astype(np.int32)).cuda()
    for alphas, thr in cases:
        if thr is None:
            post = oracle.calculate_posteriors(counts.astype(np.float64), alphas, False, 0.0)
            vals = np.unique(post[post > 0])
            pick = vals[np.linspace(0, len(vals) - 1, 12).astype(int)]
            thrs = sorted(set(float(v) for v in pick) | set(float(np.nextafter(v, 0)) for v in pick[:4]) | set(float(np.nextafter(v, 1)) for v in pick[:4]))
        else:
            thrs = [thr]
        for t in thrs:
            for keep in (False, True):
                exp = expect(counts, alphas, keep, t)
                assert np.array_equal(dev.posterior_codes_device(c16, alphas, keep, t).cpu().numpy(), exp), (alphas, t, keep)
                assert np.array_equal(dev.posterior_codes_device(c32, alphas, keep, t).cpu().numpy(), exp), (alphas, t, keep, "wide")
    # deep counts (uint32 route): beyond what a float holds exactly
    rng = np.random.default_rng(10)
    deep = (rng.integers(0, 1 << 29, size=(5000, 4)) * (rng.random((5000, 4)) < 0.6)).astype(np.int32)
    alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
    for t in (0.01, 0.24999, float(deep[7, 0] + alphas[0]) / float(deep[7].sum() + sum(alphas))):
        assert np.array_equal(dev.posterior_codes_device(torch.from_numpy(deep).cuda(), alphas, True, t).cpu().numpy(), expect(deep, alphas, True, t))


def test_connected_components_device(dev, torch_mod):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    torch = torch_mod
    rng = np.random.default_rng(4)
    n, m = 100000, 120000
    I, J = rng.integers(0, n, m).astype(np.int32), rng.integers(0, n, m).astype(np.int32)
    nc, lab = dev.connected_components_device(torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(), n)
    enc, elab = connected_components(csgraph=csr_matrix((np.ones(m), (I, J)), shape=(n, n)), directed=False)
    assert nc == enc and np.array_equal(lab.cpu().numpy(), elab)


def test_thresholded_dense_early_out(dev, oracle, torch_mod):
    """With a SNP threshold, tiles whose pairs are all past it stop reading the alignment; every pair <= threshold must
    still be exact (both encodings), and the COO output must equal the oracle's."""
    from tracs_amd import synth
    torch = torch_mod
    # long alignment: prefix pass + remainder pass over the live tiles; short one: single pass with the in-kernel early out
    # (consensus alignments are cut into site classes, csrc/site_classes.hip: the two passes run over the DENSE sites -- the
    # lineage-defining ones here, ~ 10 * mu_lin * L of them -- so the long cases are long enough for 64 stages of those)
    for n, L, mu_lin, mu_s, p_partial, enc in ((700, 400000, 3e-3, 5e-5, 0.0, "consensus"), (700, 600000, 3e-3, 5e-5, 0.0005, "general"),
                                               (700, 12000, 3e-2, 5e-4, 0.0, "consensus"), (700, 12000, 3e-2, 5e-4, 0.0005, "general")):
        # 10 well separated lineages (~ 2 * mu_lin * L = 700 SNPs apart), close samples inside (~ 2 * mu_s * L = 12)
        seqs = synth.alignment(n, L, seed=61, mu_lineage=mu_lin, mu_sample=mu_s, n_lineages=10, p_n=0.01, p_partial=p_partial)
        seqs = seqs[np.argsort(np.arange(n) % 10, kind="stable")]          # group the lineages so whole tiles are far apart
        aln = dev.Alignment(n, L)
        aln.pack(seqs)
        for thr in (40, 0):
            d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
            nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
            dev.pairsnp_dense(aln, d, nn, dist_threshold=thr)
            assert aln.encoding == enc
            er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n_threads=16)
            ri, ci = er.astype(np.int64), ec.astype(np.int64)
            dh = d.cpu().numpy().view(np.uint32)
            keep = ed <= thr
            assert np.array_equal(dh[ri[keep], ci[keep]], ed[keep].astype(np.uint32))
            assert np.array_equal(nn.cpu().numpy()[ri[keep], ci[keep]], enn[keep].astype(np.int32))
            far = dh[ri[~keep], ci[~keep]]
            assert (far.astype(np.int64) > thr).all()                      # exact, 0xFFFFFFFF or bit 31 set: never <= thr
            if L > 100000:                                                 # two-pass run: most far tiles died in the prefix pass
                cls = aln.site_classes
                assert cls is None or cls[0] >= 64 * 128 * (1 if enc == "consensus" else 2)   # the two-pass path is what this case is about
                if enc == "consensus" and not (cls and cls[2]):            # (short alignments take one plain pass: all exact)
                    assert (far >= 0x80000000).mean() > 0.5                # dead tiles are flagged
                elif enc == "consensus":                                   # minority lists add to the cells afterwards: dead tiles keep
                    stopped = far != ed[~keep].astype(np.uint32)           # a lower bound (> thr) + their list terms, never a flag
                    assert stopped.mean() > 0.5 and (far[stopped].astype(np.int64) < ed[~keep][stopped].astype(np.int64)).all()
                else:                                                      # general: dead tiles keep a lower bound that is already > thr
                    stopped = far != ed[~keep].astype(np.uint32)
                    # (256 x 128-pair workgroups: with 70-sample lineages fewer tiles are wholly far than with 128 x 128)
                    assert stopped.mean() > 0.2 and (far[stopped].astype(np.int64) < ed[~keep][stopped].astype(np.int64)).all()
            rows, cols, dd, nc = dev.coo_from_dense(d, nn, n, dist_threshold=thr)
            xr, xc, xd, xn = oracle.pairsnp_arrays(seqs, dist=thr, n_threads=16)
            assert np.array_equal(rows.cpu().numpy(), xr.astype(np.int32)) and np.array_equal(dd.cpu().numpy(), xd.astype(np.int32))
            assert np.array_equal(cols.cpu().numpy(), xc.astype(np.int32)) and np.array_equal(nc.cpu().numpy(), xn.astype(np.int32))
        aln.close()


def test_dense_source_tables_match_array_path(api, torch_mod):
    """Dense blocks take their prefix sums from (day gap, M) tables (csrc/transcluster.hip, TcTables); element arrays do not.  The
    same (N, delta) keys through both entry points -- the array path is pinned to the reference build's goldens above -- at SNP
    distances that go to the wave-per-key kernel (N >= 128) and below.  The dense keys with N >= 128 and a day gap run the term-ratio
    loop over the LINEAR tables (one exponential per lane and step: tc_eval_wave), the array path the log-space fold; a block whose
    days span 6.5 years falls back to the log-space tables (x beyond TC_LINEAR_X_MAX)."""
    from tracs_amd import device as dev
    torch = torch_mod
    n = 96
    rng = np.random.default_rng(3)
    dmat = rng.integers(0, 1500, size=(n, n)).astype(np.int32)
    dmat[:, :8] = rng.integers(0, 100, size=(n, 8))
    days = rng.integers(0, 600, size=n).astype(np.int32)
    days[:3] = days[3]                                                           # some zero gaps
    d = torch.from_numpy(dmat).cuda()
    p = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    e = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    for lamb, beta, shape in ((1e-3 * 29903, 73.0, "wide"), (5.3, 6.0, "wide"), (1e-3 * 29903, 73.0, "bench"), (1e-3 * 29903, 73.0, "long")):
        if shape == "bench":                                                     # bench.py's keys: distances near 1 000, two years of days
            dmat = rng.integers(850, 1150, size=(n, n)).astype(np.int32)
            days = rng.integers(0, 730, size=n).astype(np.int32)
            d = torch.from_numpy(dmat).cuda()
        if shape == "long":                                                      # six and a half years: x = delta (lamb + beta) beyond the linear tables
            days = rng.integers(0, 2400, size=n).astype(np.int32)
        dev.trans_dist_dense_ranges(d, n, torch.from_numpy(days).cuda(), lamb, beta, 0.01, p, e, [(0, n)], exp_p0=False)
        ii, jj = np.triu_indices(n, 1)
        N = dmat[ii, jj]
        delta = np.abs(days[ii].astype(np.int64) - days[jj].astype(np.int64)).astype(np.float64) * 86400.0 / 31556952.0
        p0, ek = api.trans_dist_arrays(N, delta, lamb, beta, 0.01)
        gp, ge = p.cpu().numpy()[ii, jj], e.cpu().numpy()[ii, jj]
        assert np.max(np.abs(gp - p0) / np.abs(p0)) < 1e-12
        fin = np.isfinite(ek) & np.isfinite(ge)
        assert fin.mean() > 0.99
        rel = np.abs(ge[fin] - ek[fin]) / np.abs(ek[fin])
        # the stopping rule can flip by a term where it is decided by rounding (tests/ek_parity.py: 'ill' keys): bounded, and rare
        assert np.quantile(rel, 0.95) < 1e-9 and rel.max() < 0.05
