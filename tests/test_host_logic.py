"""Host-side logic of the drop-in layer on CPU: command lines, CSV schema, date arithmetic, clustering
order -- against golden outputs of the REFERENCE's own Python drivers (tests/golden/python_reference_golden.json).
The GPU kernels are replaced by the oracle here (tests may use it); the GPU end-to-end run of the same
fixtures is tests/test_gpu_golden.py."""
import argparse
import json
import os
import sys

import numpy as np
import pytest


@pytest.fixture()
def pyref(golden_dir):
    with open(os.path.join(golden_dir, "python_reference_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture()
def oracle_kernels(monkeypatch, oracle):
    """Route the three kernel entry points the CLIs use through the oracle (CPU-only test harness)."""
    import tracs_amd.cluster as cl
    import tracs_amd.distance as di
    import tracs_amd.transcluster as tc
    def pairsnp_arrays(fasta, n_threads=1, dist=2147483647, filter=False):
        r, c, d, names, f, nn = oracle.pairsnp(fasta, n_threads, dist, filter)
        u = lambda x: np.asarray(x, dtype=np.uint64)
        return u(r), u(c), u(d), names, u(f), u(nn)
    # (the single-GPU command keeps its results on the device until the CSV rows -- tracs_distance_run, GPU tests; here the array
    # path, which --filter, --gpus N and incomplete metadata take, runs with the oracle behind its three kernel calls)
    monkeypatch.setenv("TRACS_DISTANCE_ARRAYS", "1")
    monkeypatch.setattr(di, "pairsnp_arrays", pairsnp_arrays)
    monkeypatch.setattr(tc, "trans_dist_arrays", oracle.trans_dist)
    monkeypatch.setattr(cl, "connected_components",
                        lambda n, I, J: (int(oracle.connected_components(n, I, J).max()) + 1 if n else 0,
                                         oracle.connected_components(n, I, J)))
    cl._ids.clear()
    yield
    cl._ids.clear()


def _materialise(pyref, td):
    from tracs_amd import synth
    d = pyref["distance"]
    seqs = np.array([np.frombuffer(s.encode(), np.uint8) for s in d["seqs"]])
    names = d["names"]
    synth.write_fasta(os.path.join(td, "refA_combined.fasta"), seqs, names=names, width=80)
    synth.write_fasta(os.path.join(td, "db.fasta"), seqs[9:], names=names[9:])
    synth.write_fasta(os.path.join(td, "query_combined.fasta.gz"), seqs[:9], names=names[:9], gz=True)
    with open(os.path.join(td, "dates.csv"), "w") as fh:
        fh.write("sample,date\n")
        for nm, iso in zip(names, d["dates"]):
            fh.write("%s,%s\n" % (nm, iso))


def _rows_equal(got, exp, float_cols=(2, 4, 5)):
    g, e = got.strip().split("\n"), exp.strip().split("\n")
    assert g[0] == e[0], "header differs"
    assert len(g) == len(e), "row count differs: %d vs %d" % (len(g), len(e))
    for a, b in zip(g[1:], e[1:]):
        fa, fb = a.split(","), b.split(",")
        assert len(fa) == len(fb) == 9
        for c in range(9):
            if c in float_cols and fa[c] != "NA":
                assert abs(float(fa[c]) - float(fb[c])) <= 1e-9 * abs(float(fb[c])) + 1e-300, (a, b)
            else:
                assert fa[c] == fb[c], (a, b)


@pytest.mark.parametrize("run", ["meta", "nometa", "meta_thr", "msadb", "filter", "filter_nometa"])
def test_distance_cli_csv_matches_reference_driver(pyref, oracle_kernels, tmp_path, monkeypatch, run):
    from tracs_amd import distance
    td = str(tmp_path)
    _materialise(pyref, td)
    r = pyref["distance"]["runs"][run]
    out = os.path.join(td, "out.csv")
    argv = [a.replace("TMP", td) for a in r["argv"]] + ["-o", out, "--loglevel", "ERROR"]
    monkeypatch.setattr(sys, "argv", ["tracs-distance"] + argv)
    distance.main()
    _rows_equal(open(out).read(), r["csv"])


def test_distance_parser_flags_and_defaults():
    from tracs_amd.distance import distance_parser
    p = distance_parser(argparse.ArgumentParser())
    a = p.parse_args(["--msa", "x.fa", "-o", "o.csv"])
    assert a.snp_threshold == 2147483647 and a.recomb_filter is False and a.metadata is None and a.msa_db is None
    assert a.clock_rate == 1e-3 * 29903 and a.trans_rate == 73.0 and a.precision == 0.01 and a.trans_threshold is None
    assert a.n_cpu == 1 and a.loglevel == "INFO" and a.msa_files == [os.path.abspath("x.fa")]
    a = p.parse_args(["--msa", "a", "b", "--msa-db", "d", "--meta", "m", "-o", "o", "-D", "5", "--filter", "--clock_rate",
                      "2.5", "--trans_rate", "9", "-K", "3", "--precision", "0.5", "-t", "8", "--loglevel", "debug"])
    assert (a.snp_threshold, a.recomb_filter, a.clock_rate, a.trans_rate, a.trans_threshold, a.precision, a.n_cpu,
            a.loglevel) == (5, True, 2.5, 9.0, 3, 0.5, 8, "DEBUG")
    for bad in (["-D", "0"], ["-t", "-1"], ["--clock_rate", "0"], ["-K", "0"]):
        with pytest.raises(SystemExit):
            p.parse_args(["--msa", "x", "-o", "o"] + bad)


def test_calculate_trans_prob_date_arithmetic(pyref, oracle_kernels):
    from datetime import date
    from tracs_amd.transcluster import calculate_trans_prob
    g = pyref["calculate_trans_prob"]
    names = pyref["distance"]["names"]
    dates = {n: (i, date.fromisoformat(i)) for n, i in zip(names, pyref["distance"]["dates"])}
    p, ek, td = calculate_trans_prob([g["rows"], g["cols"], g["d"]], sample_dates=dates, K=100, lamb=g["lamb"],
                                     beta=g["beta"], samplenames=names, log=False, precision=g["precision"])
    assert np.array_equal(td, np.array(g["time_diff"]))               # bit-identical year fractions
    days = np.array(g["days"])
    assert np.array_equal(td, np.abs(days[g["rows"]] - days[g["cols"]]).astype(np.float64) * 86400.0 / 31556952.0)
    assert np.allclose(p, g["p"], rtol=1e-9, atol=0) and np.allclose(ek, g["eK"], rtol=1e-9, atol=0)
    with pytest.raises(KeyError):                                      # a sample <= max index without a date (:27-32)
        d2 = dict(dates)
        d2.pop(names[0])
        calculate_trans_prob([g["rows"], g["cols"], g["d"]], sample_dates=d2, K=100, lamb=5.3, beta=6.0,
                             samplenames=names)


@pytest.mark.parametrize("key", ["snp_5", "direct_0.05", "expectedK_3.0", "snp_0"])
def test_cluster_cli_matches_reference_driver(pyref, oracle_kernels, tmp_path, monkeypatch, key):
    from tracs_amd import cluster
    td = str(tmp_path)
    dcsv = os.path.join(td, "d.csv")
    with open(dcsv, "w") as fh:
        fh.write(pyref["cluster"]["distance_csv"])
    col, thr = key.split("_")
    out = os.path.join(td, "c.csv")
    monkeypatch.setattr(sys, "argv", ["tracs-cluster", "-d", dcsv, "-o", out, "-c", thr, "-D", col, "--loglevel", "ERROR"])
    cluster.main()
    assert open(out).read() == pyref["cluster"]["runs"][key]


def test_cluster_ids_persist_across_calls_like_the_reference(oracle_kernels, tmp_path, monkeypatch):
    """index_count keeps its table on the function object in the reference (tracs/cluster.py:12-19)."""
    from tracs_amd import cluster
    td = str(tmp_path)
    hdr = "sampleA,sampleB,date difference,SNP distance,transmission distance,expected K,filtered SNP distance,sites considered,MSA file\n"
    a, b = os.path.join(td, "a.csv"), os.path.join(td, "b.csv")
    open(a, "w").write(hdr + "x,y,NA,1,NA,NA,0,10,r\n")
    open(b, "w").write(hdr + "z,y,NA,9,NA,NA,0,10,r\n")
    for src in (a, b):
        monkeypatch.setattr(sys, "argv", ["c", "-d", src, "-o", os.path.join(td, "o.csv"), "-c", "2", "-D", "snp",
                                          "--loglevel", "ERROR"])
        cluster.main()
    # second call: ids x=0, y=1 persist from the first file, z=2 is new; only file b's edges count (none <= 2)
    assert open(os.path.join(td, "o.csv")).read() == "sample,cluster\nx,0\ny,1\nz,2\n"


def test_main_dispatch_and_out_of_scope(monkeypatch, capsys):
    from tracs_amd import __main__ as m
    monkeypatch.setattr(sys, "argv", ["tracs", "align", "-h"])
    with pytest.raises(SystemExit):
        m.main()
    assert "not part of the MI355X distance path" in capsys.readouterr().err


def test_general_matrix_core_identity():
    """The identity behind the general matrix-core path (csrc/pairsnp_mfma.hip, csrc/general_sparse.hip): for every pair of
    IUPAC code vectors  d = L - G + 3 NN + T1 + T2  and  nn = L - c_i - c_j + NN  with the one-hot Gram G = sum |S_i n S_j|,
    NN = #(both N), T1 = sum over (partial, N) sites of |M| - 1, T2 = sum over (partial, partial) sites of (|M n M'| - 1)^+ --
    checked against the definition (src/pairsnp.hpp:398-403,417-420) on random code matrices over all 15 codes."""
    rng = np.random.default_rng(5)
    n, L = 10, 3000
    pc = np.array([bin(x).count("1") for x in range(16)])
    for p_special in (1.0, 0.05):
        codes = 1 << rng.integers(0, 4, size=(n, L))
        m = rng.random((n, L)) < p_special
        codes[m] = rng.integers(1, 16, size=int(m.sum()))
        for i in range(n):
            for j in range(i + 1, n):
                a, b = codes[i], codes[j]
                inter = pc[a & b]
                d_true, nn_true = int((inter == 0).sum()), int(((a != 15) & (b != 15)).sum())
                pa, pb = (pc[a] > 1) & (a != 15), (pc[b] > 1) & (b != 15)
                G, NN = int(inter.sum()), int(((a == 15) & (b == 15)).sum())
                T1 = int(((pc[a] - 1) * pa * (b == 15)).sum() + ((pc[b] - 1) * pb * (a == 15)).sum())
                T2 = int((np.maximum(inter - 1, 0) * (pa & pb)).sum())
                assert L - G + 3 * NN + T1 + T2 == d_true
                assert L - int((a == 15).sum()) - int((b == 15).sum()) + NN == nn_true


def test_site_class_identity():
    """The identity behind site classes (csrc/site_classes.hip): with the sites cut into variable (two samples carry different
    bases, or some sample carries a partial IUPAC code), invariant (not variable, some sample is a base) and empty (all N),
    d(i, j) over all sites = d over the variable sites, and nn(i, j) = nn over the variable sites + sum over the invariant
    sites of [i is a base][j is a base] -- checked against the definition (src/pairsnp.hpp:398-403,417-420)."""
    rng = np.random.default_rng(11)
    n, L = 12, 4000
    pc = np.array([bin(x).count("1") for x in range(16)])
    for p_partial in (0.0, 0.002):
        anc = 1 << rng.integers(0, 4, size=L)
        codes = np.tile(anc, (n, 1))
        mut = rng.random((n, L)) < 0.01
        codes[mut] = 1 << rng.integers(0, 4, size=int(mut.sum()))
        codes[rng.random((n, L)) < 0.05] = 15
        codes[:, rng.random(L) < 0.02] = 15                       # empty sites
        part = rng.random((n, L)) < p_partial
        codes[part] = rng.integers(1, 16, size=int(part.sum()))
        is_n = codes == 15
        partial = (pc[codes] > 1) & ~is_n
        seen = np.zeros((4, L), dtype=bool)                        # allele b seen at a base (non-N) position
        for b in range(4):
            seen[b] = (((codes >> b) & 1).astype(bool) & ~is_n).any(axis=0)
        var = partial.any(axis=0) | (seen.sum(axis=0) >= 2)
        inv = ~var & (~is_n).any(axis=0)
        assert 0 < var.sum() < L and inv.sum() > 0 and (~var & ~inv).sum() > 0
        for i in range(n):
            for j in range(i + 1, n):
                a, b = codes[i], codes[j]
                miss = pc[a & b] == 0
                both = ~is_n[i] & ~is_n[j]
                assert int(miss.sum()) == int(miss[var].sum())
                assert int(both.sum()) == int(both[var].sum()) + int(both[inv].sum())


def test_minority_site_identity():
    """The identity behind the minority lists (csrc/general_sparse.hip, general_fixup_kernel<MINOR>): at a site where every sample
    is N, carries exactly the reference base r, or is LISTED with allele mask M and w = [r not in M], the site's contribution to
    d(i, j) is  w_i [j carries r] + w_j [i carries r] + [M_i n M_j = {}] [both listed]  -- summed over sites in the kernel's form
        sum_{S_i} w_i + sum_{S_j} w_j - sum_{S_i} w_i [j is N] - sum_{S_j} w_j [i is N] + sum_{S_i n S_j} ([M_i n M_j = {}] - w_i - w_j)
    -- checked against the definition (src/pairsnp.hpp:398-403) on random code matrices with few listed samples per site."""
    rng = np.random.default_rng(23)
    n, L = 14, 3000
    pc = np.array([bin(x).count("1") for x in range(16)])
    ref = rng.integers(0, 4, size=L)
    codes = np.tile(1 << ref, (n, 1))
    listed = rng.random((n, L)) < 0.03
    codes[listed] = rng.integers(1, 15, size=int(listed.sum()))           # another base or a partial code (may equal {r}: not listed then)
    codes[rng.random((n, L)) < 0.05] = 15
    is_n = codes == 15
    is_listed = ~is_n & (codes != (1 << ref)[None, :])
    w = (((codes >> ref[None, :]) & 1) == 0) & is_listed
    for i in range(n):
        for j in range(i + 1, n):
            a, b = codes[i], codes[j]
            d_true = int((pc[a & b] == 0).sum())
            both = is_listed[i] & is_listed[j]
            last = (pc[a & b] == 0).astype(int) - w[i].astype(int) - w[j].astype(int)
            form = int(w[i].sum() + w[j].sum() - (w[i] & is_n[j]).sum() - (w[j] & is_n[i]).sum() + last[both].sum())
            assert form == d_true
            # what minor_fixup_kernel walks (round 5): only the listed samples with w = 1 -- two listed samples that both hold the
            # reference base add nothing to the last sum, so it is the sum over the sites at which i or j walks
            assert (last[both & ~w[i] & ~w[j]] == 0).all()
            assert int(last[both & (w[i] | w[j])].sum()) == int(last[both].sum())
