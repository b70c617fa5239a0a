"""On-disk formats either side of the distance path, host code of libtracs_hip.so, no GPU needed:
pileup text -> counts (tracs/align.py:452-473), posterior CSV (:580-596), `tracs combine` (tracs/combine.py:100-239).
Checked against oracle.py's restatements of those blocks (PARITY UNPINNED: the reference modules need pyfastx / htsbox)."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest


def _pileup_lines(rng, contigs, n_lines):
    """Lines shaped like `htsbox pileup -C -s 0` output, plus every oddity the reference's parser reacts to."""
    alleles_pool = ["A", "C", "G", "T", "N", "-1A", "+2AC", "*", "a"]
    lines = []
    for _ in range(n_lines):
        name, length = contigs[rng.integers(len(contigs))]
        pos = int(rng.integers(1, length + 1))
        refb = "ACGTNacgtR"[rng.integers(10)]
        k = int(rng.integers(1, 5))
        al = [alleles_pool[i] for i in rng.integers(0, len(alleles_pool), k)]
        if rng.random() < 0.1:
            al.append(al[0])                                  # a repeated allele: the later count wins
        fwd = [str(int(x)) for x in rng.integers(0, 40, len(al))]
        rev = [str(int(x)) for x in rng.integers(0, 40, len(al))]
        if rng.random() < 0.3:
            fwd[0] = "0"
        if rng.random() < 0.1:
            rev = rev[:-1] or ["3"]                           # ragged lists: zip() stops at the shortest
        tot = str(sum(map(int, fwd)) + sum(map(int, rev)))
        mid = ["x"] * int(rng.integers(0, 3))                 # extra columns between ref base and alleles
        sep = "\t" if rng.random() < 0.8 else "  "
        lines.append(sep.join([name, str(pos), refb] + mid + [",".join(al), tot + ":" + ",".join(fwd) + ":" + ",".join(rev)]))
    return lines


@pytest.mark.parametrize("both", [False, True])
def test_pileup_counts_match_restatement(both, oracle, tmp_path):
    from tracs_amd import align_post
    rng = np.random.default_rng(5)
    contigs = [("NC_000001.1", 700), ("plasmid|p2", 90), ("c3", 1)]
    lines = _pileup_lines(rng, contigs, 4000)
    want = oracle.pileup_counts(lines, contigs, both)
    assert want.sum() > 0 and (want[:700] > 0).any() and (want[700:790] > 0).any()
    plain = tmp_path / "p.txt"
    plain.write_text("\n".join(lines))                        # no trailing newline
    gz = tmp_path / "p.txt.gz"
    with gzip.open(gz, "wt") as f:
        f.write("\n".join(lines) + "\n")
    for path in (plain, gz):
        got = align_post.pileup_counts(str(path), contigs, both)
        assert got.dtype == np.float64 and got.shape == (791, 4)
        assert np.array_equal(got, want)


def test_pileup_counts_errors_and_edges(tmp_path):
    from tracs_amd import align_post
    contigs = [("c", 10)]

    def run(text):
        p = tmp_path / "e.txt"
        p.write_text(text)
        return align_post.pileup_counts(str(p), contigs)

    assert run("").shape == (10, 4) and run("").sum() == 0
    ok = run("c 3 A A,C 9:4,1:3,1\r\nc\t3\tA\tG\t2:1:1\n")     # CR is whitespace for str.split(); second line overwrites the row
    assert ok[2].tolist() == [0, 0, 2, 0] and ok.sum() == 2
    for bad in ("d 3 A A 1:1:0\n", "c 11 A A 1:1:0\n", "c 0 A A 1:1:0\n", "c x A A 1:1:0\n", "c 3 A A 1:x:0\n", "c 3 A A 5\n",
                "c 3\n", "c 3 A A 1:1:0\n\nc 4 A A 1:1:0\n"):
        with pytest.raises(RuntimeError, match="pileup line"):
            run(bad)
    with pytest.raises(FileNotFoundError):
        align_post.pileup_counts(str(tmp_path / "missing.txt"), contigs)


def test_read_contigs(tmp_path):
    from tracs_amd import align_post
    fa = tmp_path / "ref.fa"
    fa.write_text(">c1 some description\nACGT\nAC\n>c2\n\nGGG\n>empty\n")
    assert align_post.read_contigs(str(fa)) == [("c1", 6), ("c2", 3), ("empty", 0)]
    with gzip.open(tmp_path / "ref.fa.gz", "wt") as f:
        f.write(">x\nAC\n")
    assert align_post.read_contigs(str(tmp_path / "ref.fa.gz")) == [("x", 2)]


def test_posterior_csv_bytes(oracle, tmp_path):
    from tracs_amd import align_post
    rng = np.random.default_rng(9)
    post = rng.random((150001, 4))
    post[rng.random(post.shape) < 0.5] = 0.0
    post[:8] = [[0.000005, 0.000015, 0.999995, 1.0], [0.5, 0.125, 0.0625, 1e-7], [2.5e-6, 7.5e-6, 0.1, 0.9],
                [1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [123.456789, 0.3, 0.3, 0.3], [0.999994999, 0.5, 0.5, 0.5],
                [0.00001, 0.00002, 0.00003, 0.00004]]
    path = str(tmp_path / "post.csv.gz")
    align_post.write_posterior_csv(path, post)
    with gzip.open(path, "rb") as f:
        got = f.read()
    assert got == oracle.posterior_csv_text(post)
    align_post.write_posterior_csv(path, np.zeros((0, 4)))
    with gzip.open(path, "rb") as f:
        assert f.read() == b"\n"


def _make_align_dirs(tmp_path, rng):
    """Three sample directories as `tracs align` leaves them, two reference genomes."""
    recs = {}
    dirs = []
    for s in range(3):
        d = tmp_path / ("sample%d" % s)
        d.mkdir()
        dirs.append(str(d))
        for ref in ("GCF_1.1", "GCF_2") if s != 2 else ("GCF_1.1",):
            seq = "".join(rng.choice(list("ACGTNRYn"), 5000 + 7 * len(ref)))
            recs[("sample%d" % s, ref)] = seq
            name = "s%d_posterior_counts_ref_%s.fasta" % (s, ref)
            if s == 1:                                        # gzip + wrapped lines: any FASTA layout is accepted
                with gzip.open(d / (name + ".gz"), "wt") as f:
                    f.write(">whatever desc\n" + "\n".join(seq[i:i + 61] for i in range(0, len(seq), 61)) + "\n")
            else:
                (d / name).write_text(">s%d_%s\n%s\n" % (s, ref, seq))
        (d / ("s%d_sourmash_hits.csv" % s)).write_text(
            "intersect_bp,f_orig_query,f_match,f_unique_to_query,f_unique_weighted,average_abund,median_abund,std_abund,"
            "filename,name,md5,f_match_orig\n"
            '1000,0.5,0.25,0.125,0.1,1,1,0,db.zip,"GCF_1.1 Escherichia coli strain K-12",abc,0.3\n'
            '10,0.1,0.2,0.3,0.1,1,1,0,db.zip,"GCF_9 Unmapped species",abc,0.3\n')
    return dirs, recs


def test_combine_matches_restatement(oracle, hiplib, tmp_path):
    from tracs_amd import combine
    rng = np.random.default_rng(21)
    dirs, recs = _make_align_dirs(tmp_path, rng)
    out = str(tmp_path / "combined")
    rc = subprocess.run([sys.executable, "-m", "tracs_amd", "combine", "-i"] + dirs + ["-o", out, "-t", "3"],
                        capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert rc.returncode == 0, rc.stderr
    for ref in ("GCF_1.1", "GCF_2"):
        samples = [s for s in ("sample0", "sample1", "sample2") if (s, ref) in recs]
        text, ncov = oracle.combined_fasta_text([(s, recs[(s, ref)]) for s in samples])
        path = os.path.join(out, ref + "_combined.fasta.gz")
        with gzip.open(path, "rt") as f:                      # one gzip member per sample reads as one stream
            assert f.read() == text
        # the file is what `tracs distance` ingests: our reader sees the same records
        import ctypes as C
        n, L = C.c_size_t(0), C.c_size_t(0)
        assert hiplib.tracs_debug_read_fasta(path.encode(), C.byref(n), C.byref(L), None) == 0
        assert n.value == len(samples) and L.value == len(recs[(samples[0], ref)])
        got = combine.write_alignment(ref, [(s, os.path.join(tmp_path, s, p)) for s in samples
                                            for p in os.listdir(tmp_path / s) if "_ref_" + ref + ".fasta" in p],
                                      os.path.join(out, "again_"))
        assert got == {(s, ref): ncov[s] for s in samples}
    meta = open(os.path.join(out, "combined_metadata.csv")).read().splitlines()
    assert meta[0] == ("sample,accession,intersect_bp,f_orig_query,f_match,f_unique_to_query,coverage,mean_depth,"
                       "mean_nonzero_depth,frac_N,species")
    s0 = recs[("sample0", "GCF_1.1")]
    assert meta[1] == "sample0,GCF_1.1,1000,0.5,0.25,0.125,NA,NA,NA,%s,Escherichia coli strain K-12" % str(s0.count("N") / len(s0))
    assert meta[2] == "sample0,GCF_9,10,0.1,0.2,0.3,NA,NA,NA,NA,Unmapped species"
    assert len(meta) == 1 + 2 * 3


def test_combine_refuses_multi_record_and_bad_names(tmp_path):
    from tracs_amd import combine
    d = tmp_path / "s"
    d.mkdir()
    (d / "s_posterior_counts_ref_X.fasta").write_text(">a\nACGT\n>b\nACGT\n")
    with pytest.raises(SystemExit):
        combine.write_alignment("X", [("s", str(d / "s_posterior_counts_ref_X.fasta"))], str(tmp_path) + os.sep)
    with pytest.raises(SystemExit):
        combine.find_ref("not_an_align_output.txt")
    assert combine.find_ref("/x/y/p_posterior_counts_ref_GCF_000005845.2.fasta.gz") == "GCF_000005845.2"


def _tr_member(text):
    """One gzip member with the "TR" size subfield, as tracs_combine_fasta writes them."""
    import struct
    import zlib
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(text) + co.flush()
    total = 10 + 2 + 12 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\x03" + struct.pack("<H", 12) + b"TR" + struct.pack("<HQ", 8, total) + body +
            struct.pack("<II", zlib.crc32(text), len(text) & 0xFFFFFFFF))


def test_indexed_member_reader_equals_serial_state_machine(hiplib, tmp_path):
    """The parallel reader for combine's own files (one gzip member per sample, sizes in an FEXTRA subfield) must give
    exactly what the serial kseq restatement gives, and must step aside for anything that is not one clean record per
    member -- the serial reader then produces the reference's behaviour and messages."""
    import ctypes as C
    from tracs_amd import combine

    def read(path, serial):
        if serial:
            os.environ["TRACS_SERIAL_FASTA"] = "1"
        n, L, h = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        rc = hiplib.tracs_debug_read_fasta(os.fsencode(path), C.byref(n), C.byref(L), C.byref(h))
        os.environ.pop("TRACS_SERIAL_FASTA", None)
        return rc, n.value, L.value, h.value, hiplib.tracs_last_error()

    rng = np.random.default_rng(4)
    alns = []
    for s in range(37):
        d = tmp_path / ("d%d" % s)
        d.mkdir()
        seq = "".join(rng.choice(list("ACGTNacgtRY-"), 20011))
        (d / "x_posterior_counts_ref_R.fasta").write_text(">x%d something\n%s\n" % (s, seq))
        alns.append(("sample_%d" % s, str(d / "x_posterior_counts_ref_R.fasta")))
    combine.write_alignment("R", alns, str(tmp_path) + os.sep, n_threads=5)
    good = str(tmp_path / "R_combined.fasta.gz")
    raw = open(good, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"TR"
    fast, slow = read(good, False), read(good, True)
    assert fast == slow and fast[0] == 0 and fast[1] == 37 and fast[2] == 20011
    with gzip.open(good, "rb") as f:                                        # and it is an ordinary gzip file
        assert f.read().count(b">") == 37

    cases = {
        "two_records_in_a_member": _tr_member(b">a\nACGT\n") + _tr_member(b">b\nACGT\n>c\nACGT\n"),
        "ragged": _tr_member(b">a\nACGT\n") + _tr_member(b">b\nACG\n"),
        "fastq": _tr_member(b"@a\nACGT\n+\nIIII\n") + _tr_member(b"@b\nACGT\n+\nIIII\n"),
        "junk_before_header": _tr_member(b"xx>a\nACGT\n") + _tr_member(b">b\nACGT\n"),
        "chain_then_plain_member": _tr_member(b">a\nACGT\n") + gzip.compress(b">b\nACGT\n"),
        "wrapped_lines_and_crlf": _tr_member(b">a x\r\nAC\r\nGT\r\n") + _tr_member(b">b\nA\nC\nG\nT"),
        "truncated_size": _tr_member(b">a\nACGT\n")[:-3],
    }
    for name, blob in cases.items():
        path = str(tmp_path / (name + ".fa.gz"))
        open(path, "wb").write(blob)
        assert read(path, False) == read(path, True), name
    assert read(str(tmp_path / "two_records_in_a_member.fa.gz"), False)[1] == 3
    assert read(str(tmp_path / "ragged.fa.gz"), False)[0] != 0
    assert read(str(tmp_path / "wrapped_lines_and_crlf.fa.gz"), False)[1:3] == (2, 4)


def test_distance_rows_writer_is_python_str_exact(hiplib, tmp_path):
    """tracs_write_distance_rows == the reference's ",".join([... str(x) ...]) + "\\n" loop (tracs/distance.py:212-258):
    shortest-repr floats in Python's notation, the -K filter, and the two "NA" layouts."""
    from tracs_amd import distance
    rng = np.random.default_rng(12)
    n, ns = 70001, 300
    names = ["s%d|x y" % i if i % 7 else "Sample-%d" % i for i in range(ns)]
    rows = rng.integers(0, ns, n).astype(np.uint64)
    cols = rng.integers(0, ns, n).astype(np.uint64)
    snpd = rng.integers(0, 3000, n).astype(np.uint64)
    filt = rng.integers(0, 3000, n).astype(np.uint64)
    nn = rng.integers(0, 5000000, n).astype(np.uint64)
    ddiff = rng.integers(0, 730, n) * 86400.0 / 31556952.0
    tdist = np.exp(rng.uniform(-60, 0, n))
    tdist[:5] = [0.0, 1.0, 1e-5, 9.999e-5, 1e-320]
    ek = np.exp(rng.uniform(-12, 40, n))
    ek[5:9] = [np.nan, np.inf, 5.0, 1e16]

    def expect(with_dates, filt_col, kmax):
        out = []
        for t in range(n):
            if with_dates:
                if kmax is None or kmax >= ek[t]:
                    out.append(",".join([names[rows[t]], names[cols[t]], str(ddiff[t]), str(int(snpd[t])), str(tdist[t]), str(ek[t]),
                                         "NA" if filt_col is None else str(int(filt_col[t])), str(int(nn[t])), "REF"]) + "\n")
            else:
                out.append(",".join([names[rows[t]], names[cols[t]], "NA", str(int(snpd[t])), "NA", "NA", str(int(filt_col[t])),
                                     str(int(nn[t])), "REF"]) + "\n")
        return "".join(out)

    for with_dates, fc, kmax in ((True, None, None), (True, filt, 5), (True, None, 0), (False, filt, None)):
        path = str(tmp_path / "rows.csv")
        open(path, "w").write("HEADER\n")
        w = distance._append_rows(path, names, rows, cols, snpd, fc, nn, ddiff if with_dates else None, tdist if with_dates else None,
                                  ek if with_dates else None, kmax, "REF")
        want = expect(with_dates, fc, kmax)
        assert open(path).read() == "HEADER\n" + want
        assert w == want.count("\n")


def test_distance_edge_reader_equals_python_scan(hiplib, tmp_path):
    """tracs_read_distance_edges vs the reference's own loop (tracs/cluster.py:100-116) on a multi-megabyte CSV that is
    parsed in many chunks: ids by first appearance in FILE order, every column, nan/inf/padded values, error cases."""
    from tracs_amd import cluster as cl
    rng = np.random.default_rng(31)
    ns, n = 900, 120000
    names = ["iso_%d" % i for i in rng.permutation(ns)]
    lines = ["sampleA,sampleB,date difference,SNP distance,transmission distance,expected K,filtered SNP distance,sites considered,MSA file"]
    for t in range(n):
        a, b = rng.integers(0, ns, 2)
        ek = rng.choice(["nan", "inf", " 3.5 ", "1e-05", str(float(np.exp(rng.uniform(-5, 8))))])
        lines.append("%s,%s,%s,%d,%s,%s,%d,%d,ref" % (names[a], names[b], str(rng.random()), rng.integers(0, 200), str(float(rng.random())),
                                                     ek, rng.integers(0, 200), rng.integers(0, 10**6)))
    path = str(tmp_path / "dist.csv")
    open(path, "w").write("\n".join(lines) + ("\n" if n % 2 else ""))
    assert os.path.getsize(path) > 8 << 20

    def python_scan(col, thr, ids):
        I, J, count = [], [], 0
        for line in lines[1:]:
            f = line.strip().split(",")
            for nm in (f[0], f[1]):
                if nm not in ids:
                    ids[nm] = len(ids)
            if float(f[col]) <= thr:
                I.append(ids[f[0]])
                J.append(ids[f[1]])
            count += 1
        return count, I, J

    for col, thr in ((3, 20.0), (5, 4.0), (4, 0.5), (6, 1e9)):
        cl._ids.clear()
        cl._ids["seeded_before"] = 0                                 # ids persist across calls in one process
        want_ids = {"seeded_before": 0}
        wc, wi, wj = python_scan(col, thr, want_ids)
        gc, gi, gj = cl.read_edges(path, col, thr)
        assert gc == wc and gi.tolist() == wi and gj.tolist() == wj
        assert list(cl._ids.items()) == list(want_ids.items())
    cl._ids.clear()
    for bad, exc in (("h\na,b,1,NA,0.1,2,3,4,r\n", ValueError), ("h\na,b,1\n", IndexError), ("h\na,b,1,2,3,4,5,6,r\n\n", IndexError),
                     ("", StopIteration)):
        p2 = str(tmp_path / "bad.csv")
        open(p2, "w").write(bad)
        with pytest.raises(exc):
            cl.read_edges(p2, 3, 1.0)
        cl._ids.clear()
    open(p2, "w").write("header only\n")
    assert cl.read_edges(p2, 3, 1.0)[0] == 0
    with pytest.raises(FileNotFoundError):
        cl.read_edges(str(tmp_path / "nope.csv"), 3, 1.0)
