#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the build container only).

Sources of truth, in order of authority:
  * oracle/_ref  -- the reference's own src/transcluster.hpp + src/dmultinomial.hpp + src/kseq.h compiled
    from where they lie under /root/reference with setup.py's flags (oracle/Makefile);
  * the reference's Python drivers imported from /root/reference (tracs/transcluster.py, tracs/cluster.py,
    tracs/distance.py, tracs/dirichlet_multinomial.py) with a stub `TRACS` module whose trans_dist /
    calculate_posteriors / lprob_k_given_N are oracle/_ref and whose pairsnp is OUR oracle
    (src/pairsnp.hpp cannot be built here: Boost is absent -- pairsnp fixtures are therefore
    "unpinned" and additionally cross-checked against an independent numpy brute force).
Only DATA is written (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import oracle as O  # noqa: E402
from tracs_amd import synth  # noqa: E402

R = O.ref_module()
assert R is not None, "oracle/_ref is not built (make -C oracle)"


def jdump(name, obj):
    with open(os.path.join(HERE, name), "w") as fh:
        json.dump(obj, fh, indent=1)
    print("wrote", name)


# ---------------------------------------------------------------------------------------
def transcluster():
    rng = np.random.default_rng(20241022)
    out = {"known_answers": {
        # /root/reference/tests/test_llk.py:21-29
        "lprob_k_given_N": {"args": [7, 4, 0.16963, 3, 52], "lgamma_len": 20,
                            "expect": [-17.9565184209608, 12.0861694243766], "atol": 1e-6},
        # /root/reference/tests/test_trans_distance.py:29-42 (1-day gap; SNP 0 and 2; CLI defaults)
        "trans_distance": {"snp": [0, 2], "delta": [0.002737907006988508] * 2, "lamb": 1e-3 * 29903, "beta": 73.0,
                           "precision": 0.01, "p_direct": [0.23794988406662973, 0.024467137572328577],
                           "expected_k": [2.6335200453700187, 7.315670110063259], "atol": 1e-6}}}
    grids = []
    for lamb, beta, thr in ((1e-3 * 29903, 73.0, 0.01), (5.3, 6.0, 0.01), (3.0, 52.0, 0.01), (5.3, 6.0, 1e-4),
                            (20.0, 2.0, 0.05)):
        N = np.concatenate([np.arange(0, 12), rng.integers(0, 120, 60)]).astype(int)
        days = np.concatenate([np.zeros(6, int), rng.integers(1, 700, N.size - 6)])
        delta = days.astype(np.float64) * 86400.0 / 31556952.0
        p0, ek = R.ref_trans_dist(N.tolist(), delta.tolist(), lamb, beta, thr)
        cls = [O.ek_conditioning(int(n), float(d), lamb, beta, thr)[0] for n, d in zip(N, delta)]
        grids.append({"lamb": lamb, "beta": beta, "thr": thr, "N": N.tolist(), "days": days.tolist(),
                      "delta": delta.tolist(), "p0": list(p0), "eK": list(ek), "conditioning": cls})
    out["trans_dist"] = grids
    lg = [float(x) for x in __import__("scipy.special", fromlist=["gammaln"]).gammaln(np.arange(300))]
    rows = []
    for _ in range(60):
        N, k = int(rng.integers(0, 120)), int(rng.integers(0, 120))
        delta = float(rng.choice([0.0, 0.0027379070069885, 0.05, 0.4, 1.7]))
        lamb, beta = (5.3, 6.0) if rng.random() < 0.5 else (29.903, 73.0)
        a = R.ref_lprob_k_given_N(N, k, delta, lamb, beta, lg)
        b = R.ref_lprob_k_given_N_2(N, k, delta, lamb, beta)
        rows.append({"N": N, "k": k, "delta": delta, "lamb": lamb, "beta": beta, "lprob_k_given_N": list(a),
                     "lprob_k_given_N_2": list(b)})
    out["lprob"] = {"lgamma_len": 300, "rows": rows}
    jdump("transcluster_golden.json", out)


def transcluster_large_n():
    """Keys with hundreds to thousands of SNPs (the bench workload's range: mean d ~ 1 000) through oracle/_ref: these take the
    device's wave-per-key kernel from their first term (csrc/transcluster.hip, TC_WAVE_PREFIX_MIN).  A file of its own so the
    older fixtures stay byte-identical."""
    rng = np.random.default_rng(20261002)
    grids = []
    for lamb, beta, thr in ((1e-3 * 29903, 73.0, 0.01), (5.3, 6.0, 0.01)):
        N = np.concatenate([[128, 129, 191, 192, 193, 1000], rng.integers(128, 1600, 40)]).astype(int)
        days = np.concatenate([[1, 30, 365, 700, 2, 3], rng.integers(1, 700, N.size - 6)])
        delta = days.astype(np.float64) * 86400.0 / 31556952.0
        p0, ek = R.ref_trans_dist(N.tolist(), delta.tolist(), lamb, beta, thr)
        cls = [O.ek_conditioning(int(n), float(d), lamb, beta, thr)[0] for n, d in zip(N, delta)]
        grids.append({"lamb": lamb, "beta": beta, "thr": thr, "N": N.tolist(), "days": days.tolist(),
                      "delta": delta.tolist(), "p0": list(p0), "eK": list(ek), "conditioning": cls})
    jdump("transcluster_golden_large_n.json", {"trans_dist": grids})


def transcluster_outbreak():
    """4 000 outbreak-scale keys at the CLI defaults (N <= 80 SNPs, 1..730 days, lamb = 29.903, beta = 73, precision 0.01) -- the
    regime TRACS is used in, where a few percent of the keys have their E(K) stopping point decided by rounding ('ill').  Three
    columns: the reference as shipped (oracle/_ref: setup.py's -ffast-math), the SAME source compiled IEEE-strict
    (oracle/_ref/_tracs_ref_strict), and the oracle's conditioning class.  Each build runs in its own child process (loading the
    fast-math library switches a process to flush-to-zero)."""
    rng = np.random.default_rng(20261003)
    n_keys = 4000
    N = rng.integers(0, 81, n_keys)
    days = rng.integers(1, 731, n_keys)
    lamb, beta, thr = 1e-3 * 29903, 73.0, 0.01
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "keys.npz")
        np.savez(f, N=N, days=days)
        code = ("import sys, json, numpy as np\n"
                "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import importlib; M = importlib.import_module(sys.argv[2])\n"
                "z = np.load(sys.argv[1]); N = z['N'].tolist(); d = (z['days'].astype(np.float64) * 86400.0 / 31556952.0).tolist()\n"
                "p0, ek = M.ref_trans_dist(N, d, %r, %r, %r)\n"
                "out = {'p0': list(p0), 'eK': list(ek)}\n"
                "if sys.argv[2].endswith('strict'):\n"
                "    from oracle import oracle as O\n"
                "    out['cls'] = [O.ek_conditioning(int(n), float(x), %r, %r, %r)[0] for n, x in zip(N, d)]\n"
                "print(json.dumps(out))\n" % (ROOT, os.path.join(ROOT, "oracle", "_ref"), lamb, beta, thr, lamb, beta, thr))
        res = {}
        for mod in ("_tracs_ref", "_tracs_ref_strict"):
            o = subprocess.run([sys.executable, "-c", code, f, mod], capture_output=True, text=True, timeout=7200)
            assert o.returncode == 0, o.stderr[-2000:]
            res[mod] = json.loads(o.stdout.strip().splitlines()[-1])
    fast, strict = res["_tracs_ref"], res["_tracs_ref_strict"]

    def clean(v):
        return [x if np.isfinite(x) else None for x in v]
    jdump("transcluster_outbreak_golden.json",
          {"lamb": lamb, "beta": beta, "thr": thr, "N": N.tolist(), "days": days.tolist(),
           "p0": fast["p0"], "eK": clean(fast["eK"]), "eK_strict_build": clean(strict["eK"]), "conditioning": strict["cls"],
           "note": "eK: the reference as shipped (-ffast-math); eK_strict_build: the same headers compiled without it; null = inf / nan"})


def posteriors():
    counts = synth.allele_counts(4000, seed=11, depth=20, p_two=0.08).astype(np.float64)
    counts[:40] = 0
    counts[40:80] = 5
    counts[80:120, 2] = counts[80:120, 0]
    counts[120:160, 3] = counts[120:160, 1]
    counts[160:170] = [[3, 3, 1, 1]] * 10
    cases = []
    for alphas in ([20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1], [0.5, 12.0, 0.05, 3.0], [0, 0, 0, 1.0]):
        for keep in (False, True):
            for thr in (0.0, 0.01, 0.1):
                post = R.ref_calculate_posteriors(counts, list(alphas), keep, thr)
                cases.append({"alphas": alphas, "keep": keep, "threshold": thr, "posterior": np.asarray(post)})
    np.savez_compressed(os.path.join(HERE, "posteriors_golden.npz"), counts=counts,
                        meta=json.dumps([{k: c[k] for k in ("alphas", "keep", "threshold")} for c in cases]),
                        **{"post_%d" % i: c["posterior"] for i, c in enumerate(cases)})
    print("wrote posteriors_golden.npz")


KSEQ_CASES = {
    "plain": ">s1\nACGT\n>s2 desc text\nAC\nGT\n",
    "wrapped_crlf": ">a\r\nAC\r\nGT\r\n>b\r\nTTGA\r\n",
    "leading_junk": "junk line\n\n>x\nACGTN\n",
    "no_trailing_newline": ">x\nACGT\n>y\nAC-T",
    "lower_and_iupac": ">m\nacgtRYKMswbdhvn-?.*\n",
    "fastq": "@r1 c\nACGT\n+\nIIII\n@r2\nTTGA\n+r2\n!!!!\n",
    "fastq_short_qual": "@r1\nACGT\n+\nII\n",
    "gt_inside_sequence": ">a\nAC>b\nGT\n",
    "tabs_and_spaces": ">name\twith tab\nA C\tG T\n",
    "empty_name": ">\nACGT\n",
    "empty_file": "",
    "header_only": ">lonely",
    "at_header_fasta_body": "@odd\nACGT\n>next\nTTTT\n",
    "blank_lines": ">a\n\nAC\n\nGT\n\n>b\nAAAA\n",
}


def kseq():
    dump = O.kseq_dump_path()
    assert dump
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, text in KSEQ_CASES.items():
            p = os.path.join(td, name)
            with open(p, "wb") as fh:
                fh.write(text.encode("latin-1"))
            pr = subprocess.run([dump, p], capture_output=True)
            if pr.returncode != 0:          # the reference's reader itself crashes (e.g. a header with no sequence bytes
                out[name] = {"text": text, "crash": True}      # dereferences a NULL kstring): behaviour undefined, not a fixture
                continue
            res = pr.stdout.decode("latin-1")
            lines = res.split("\n")
            rc = int([ln for ln in lines if ln.startswith("#rc=")][0][4:])
            recs = [ln.split("\t") for ln in lines if ln and not ln.startswith("#rc=")]
            out[name] = {"text": text, "rc": rc, "records": [[r[0], r[1] if len(r) > 1 else ""] for r in recs]}
    jdump("kseq_golden.json", out)


def pairsnp_unpinned():
    cases = {}
    specs = {"iupac_small": dict(n=7, L=61, seed=3, mu_lineage=0.05, mu_sample=0.03, p_n=0.05, p_partial=0.1, p_lower=0.2, p_other=0.05),
             "tail_129": dict(n=12, L=129, seed=4, mu_lineage=0.02, mu_sample=0.02, p_n=0.02, p_partial=0.02),
             "consensus_300": dict(n=20, L=300, seed=5, mu_lineage=0.01, mu_sample=0.01, p_n=0.03)}
    for name, kw in specs.items():
        seqs = synth.alignment(**kw)
        for mode, n0, dist in (("all", None, 2147483647), ("thr", None, 4), ("twofile", kw["n"] // 3, 2147483647)):
            a = O.pairsnp_arrays(seqs, n0=n0, dist=dist)
            b = O.brute_pairsnp(seqs, n0=n0, dist=dist)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), "oracle != brute force"
            cases["%s/%s" % (name, mode)] = {"seqs": [row.tobytes().decode("ascii") for row in seqs], "n0": n0, "dist": dist,
                                             "rows": a[0].tolist(), "cols": a[1].tolist(), "d": a[2].tolist(),
                                             "nn": a[3].tolist()}
    jdump("pairsnp_unpinned.json", {"status": "PARITY UNPINNED: produced by oracle/tracs_oracle.c (reference src/pairsnp.hpp is "
                                              "unbuildable here: Boost absent), cross-checked against oracle.brute_pairsnp",
                                    "cases": cases})


def python_reference():
    """Golden outputs of the reference's Python drivers (imported from /root/reference)."""
    stub = types.ModuleType("TRACS")
    stub.pairsnp = O.pairsnp
    stub.trans_dist = R.ref_trans_dist
    stub.calculate_posteriors = R.ref_calculate_posteriors
    stub.lprob_k_given_N = R.ref_lprob_k_given_N
    sys.modules["TRACS"] = stub
    sys.modules.setdefault("pyfastx", types.ModuleType("pyfastx"))
    sys.path.insert(0, REF)
    import importlib
    ref_distance = importlib.import_module("tracs.distance")
    ref_cluster = importlib.import_module("tracs.cluster")
    ref_tc = importlib.import_module("tracs.transcluster")
    ref_dm = importlib.import_module("tracs.dirichlet_multinomial")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        seqs = synth.alignment(14, 900, seed=21, mu_lineage=0.004, mu_sample=0.002, p_n=0.02, p_partial=0.01)
        names = ["iso%02d" % i for i in range(14)]
        msa = os.path.join(td, "refA_combined.fasta")
        synth.write_fasta(msa, seqs, names=names, width=80)
        iso, days = synth.dates(14, seed=21, span_days=120)
        meta = os.path.join(td, "dates.csv")
        with open(meta, "w") as fh:
            fh.write("sample,date\n")
            for nm, d in zip(names, iso):
                fh.write("%s,%s\n" % (nm, d))
        db = os.path.join(td, "db.fasta")
        synth.write_fasta(db, seqs[9:], names=names[9:])
        q = os.path.join(td, "query_combined.fasta.gz")
        synth.write_fasta(q, seqs[:9], names=names[:9], gz=True)
        runs = {"meta": ["--msa", msa, "--meta", meta], "nometa": ["--msa", msa],
                "meta_thr": ["--msa", msa, "--meta", meta, "-D", "6", "-K", "12", "--clock_rate", "5.3", "--trans_rate", "6.0"],
                "msadb": ["--msa", q, "--msa-db", db, "--meta", meta, "-D", "40"],
                "filter": ["--msa", msa, "--meta", meta, "--filter", "--clock_rate", "5.3", "--trans_rate", "6.0"],
                "filter_nometa": ["--msa", msa, "--filter", "-D", "30"]}
        csvs = {}
        for key, argv in runs.items():
            o = os.path.join(td, key + ".csv")
            sys.argv = [""] + argv + ["-o", o, "--loglevel", "ERROR"]
            ref_distance.main()
            csvs[key] = open(o).read().replace(td, "TMP")
        out["distance"] = {"seqs": [r.tobytes().decode() for r in seqs], "names": names, "dates": iso,
                           "runs": {k: {"argv": [a.replace(td, "TMP") for a in v], "csv": csvs[k]} for k, v in runs.items()}}
        # calculate_trans_prob (dates -> delta bits)
        rr, cc, dd, nm, _, _ = O.pairsnp([msa], 1, 2147483647, False)
        dates = {n: (i, __import__("datetime").date.fromisoformat(i)) for n, i in zip(names, iso)}
        p0, ek, td_ = ref_tc.calculate_trans_prob([rr, cc, dd], sample_dates=dates, K=100, lamb=5.3, beta=6.0,
                                                  samplenames=nm, log=False, precision=0.01)
        out["calculate_trans_prob"] = {"rows": rr, "cols": cc, "d": dd, "days": days.tolist(), "lamb": 5.3, "beta": 6.0,
                                       "precision": 0.01, "p": list(map(float, p0)), "eK": list(map(float, ek)),
                                       "time_diff": list(map(float, td_))}
        # cluster: labels for each -D column
        clus = {}
        for col, thr in (("snp", 5), ("direct", 0.05), ("expectedK", 3.0), ("snp", 0)):
            o = os.path.join(td, "clu.csv")
            dist_csv = os.path.join(td, "meta.csv")
            ref_cluster.index_count.__dict__.pop("dict", None)
            ref_cluster.index_count.__dict__.pop("curr", None)
            sys.argv = ["", "-d", dist_csv, "-o", o, "-c", str(thr), "-D", col, "--loglevel", "ERROR"]
            ref_cluster.main()
            clus["%s_%s" % (col, thr)] = open(o).read()
        out["cluster"] = {"distance_csv": csvs["meta"], "runs": clus}
    # find_dirichlet_priors known answer + extra cases
    import contextlib
    import io
    cnt = np.array([[1, 19, 73], [1, 19, 90], [0, 33, 53], [5, 19, 91], [3, 17, 57], [3, 13, 77], [5, 6, 89], [1, 23, 85],
                    [2, 29, 67], [7, 6, 99], [0, 17, 96], [0, 10, 86], [4, 5, 85], [6, 25, 65], [0, 5, 86], [0, 16, 91],
                    [23, 14, 73], [4, 9, 96], [2, 19, 71], [9, 24, 78]])          # /root/reference/tests/test_dirichlet_multinomial.py:12
    with contextlib.redirect_stdout(io.StringIO()):
        fp = ref_dm.find_dirichlet_priors(cnt.astype(float), tol=1e-10, method="FP")
        loo = ref_dm.find_dirichlet_priors(cnt.astype(float), tol=1e-10, method="LOO")
        c4 = synth.allele_counts(3000, seed=9, depth=25, p_two=0.06).astype(float)
        fp4 = ref_dm.find_dirichlet_priors(c4, method="FPI", error_filt_threshold=0.01)
        fp4b = ref_dm.find_dirichlet_priors(c4[:40], method="FPI")
    out["find_dirichlet_priors"] = {"r_mglm": [20.8156311152126, 4.38181182238621, 0.889048781117318],
                                    "counts3": cnt.tolist(), "fp": list(map(float, fp)), "loo": list(map(float, loo)),
                                    "counts4_seed": 9, "fp4_filt0.01": list(map(float, fp4)), "fp4_first40": list(map(float, fp4b))}
    jdump("python_reference_golden.json", out)


if __name__ == "__main__":
    if sys.argv[1:] == ["outbreak"]:             # only the fixture added in round 3
        transcluster_outbreak()
        sys.exit(0)
    if sys.argv[1:] == ["large-n"]:              # only the fixture added in round 2
        transcluster_large_n()
        sys.exit(0)
    transcluster()
    transcluster_large_n()
    posteriors()
    kseq()
    pairsnp_unpinned()
    python_reference()
