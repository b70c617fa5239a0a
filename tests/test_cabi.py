"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol include/tracs_hip.h
declares.  No compute calls here (no GPU in this container) -- only entry points that do not touch the device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tracs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tracs_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    from tracs_amd import _lib
    assert sorted(_lib.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol(hiplib):
    for name in _declared():
        assert hasattr(hiplib, name), "libtracs_hip.so does not export " + name
    assert hiplib.tracs_abi_version() == 1


def test_built_for_gfx950():
    from tracs_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"pairsnp_tile_kernel" in blob and b"tc_keys_kernel" in blob


def test_iupac_table_matches_oracle(hiplib, oracle):
    O = oracle.lib()
    for ch in range(256):
        assert hiplib.tracs_debug_iupac_mask(ch) == O.orc_iupac_mask(ch), ch


def test_argument_errors_do_not_need_a_gpu(hiplib, tmp_path):
    h = C.c_void_p()
    paths = (C.c_char_p * 3)(b"a", b"b", b"c")
    rc = hiplib.tracs_pairsnp(paths, 3, 1, 10, 0, C.byref(h))
    assert rc == -1 and hiplib.tracs_last_error() == b"Invalid number of fasta files!"      # src/pairsnp.hpp:340-343
    ragged = os.path.join(str(tmp_path), "r.fa")
    with open(ragged, "w") as fh:
        fh.write(">a\nACGT\n>b\nAC\n")
    one = (C.c_char_p * 1)(ragged.encode())
    rc = hiplib.tracs_pairsnp(one, 1, 1, 10, 0, C.byref(h))
    assert rc == -4 and hiplib.tracs_last_error() == b"Error reading FASTA, variable sequence lengths!"   # :94-98
    bad = os.path.join(str(tmp_path), "q.fq")
    with open(bad, "w") as fh:
        fh.write("@r\nACGT\n+\nII\n")
    one = (C.c_char_p * 1)(bad.encode())
    rc = hiplib.tracs_pairsnp(one, 1, 1, 10, 0, C.byref(h))
    assert rc == -2 and hiplib.tracs_last_error() == b"Error reading FASTA!"                 # :84-91
    missing = (C.c_char_p * 1)(os.path.join(str(tmp_path), "nope.fa").encode())
    assert hiplib.tracs_pairsnp(missing, 1, 1, 10, 0, C.byref(h)) == -5


def test_no_cpu_fallback_without_a_gpu(hiplib, tmp_path):
    """On a box without a GPU the product refuses to compute; it never falls back to the oracle."""
    if hiplib.tracs_device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np
    from tracs_amd import _lib, api
    fa = os.path.join(str(tmp_path), "a.fa")
    with open(fa, "w") as fh:
        fh.write(">a\nACGT\n>b\nACGA\n")
    with pytest.raises(_lib.TracsError, match="no CPU fallback"):
        api.pairsnp(fasta=[fa], n_threads=1, dist=10, filter=False)
    with pytest.raises(_lib.TracsError, match="no CPU fallback"):
        api.trans_dist([1], [0.1], 5.3, 6.0, 0.01)
    with pytest.raises(_lib.TracsError, match="no CPU fallback"):
        api.calculate_posteriors(np.ones((3, 4)), [1, 1, 1, 1], False, 0.01)
    with pytest.raises(_lib.TracsError, match="no CPU fallback"):
        api.lprob_k_given_N(1, 1, 0.1, 5.3, 6.0, [0.0] * 10)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tracs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text and "oracle/" not in text.replace("oracle/.", ""), f
    assert "oracle" not in open(os.path.join(ROOT, "TRACS.py")).read()


def test_parallel_fasta_reader_equals_serial_state_machine(hiplib, tmp_path):
    """Large well-formed plain FASTA takes the multi-threaded mmap path; it must return exactly what the serial kseq
    restatement returns, and anything not well-formed must fall back (same results, same errors)."""
    import numpy as np

    def parse(path, serial):
        if serial:
            os.environ["TRACS_SERIAL_FASTA"] = "1"
        else:
            os.environ.pop("TRACS_SERIAL_FASTA", None)
        n, L, h = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        rc = hiplib.tracs_debug_read_fasta(path.encode(), C.byref(n), C.byref(L), C.byref(h))
        os.environ.pop("TRACS_SERIAL_FASTA", None)
        return rc, n.value, L.value, h.value, hiplib.tracs_last_error()

    rng = np.random.default_rng(0)
    L, n = 1_000_003, 80
    base = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, L)]
    good = os.path.join(str(tmp_path), "big.fa")
    with open(good, "wb") as f:
        for s in range(n):
            row = base.copy()
            row[rng.integers(0, L, 100)] = ord("a")
            sep = b"\r\n" if s % 7 == 3 else b"\n"
            f.write(b">rec%d some comment > with gt\n" % s)
            f.write(sep.join(row[o:o + 70].tobytes() for o in range(0, L, 70)))
            if s != n - 1:
                f.write(sep)
    a, b = parse(good, False), parse(good, True)
    assert a[0] == 0 and a[:4] == b[:4] and a[1] == n and a[2] == L
    # a stray '>' inside a sequence line: kseq splits the record there -> ragged; both paths must say so
    bad = os.path.join(str(tmp_path), "bad.fa")
    data = open(good, "rb").read()
    k = data.index(b"\n", 200) + 30
    open(bad, "wb").write(data[:k] + b">" + data[k + 1:])
    a, b = parse(bad, False), parse(bad, True)
    assert a[0] == b[0] == -4 and a[4] == b[4] == b"Error reading FASTA, variable sequence lengths!"
    # junk before the first header: the serial rules skip it; the fast path must decline and agree
    junk = os.path.join(str(tmp_path), "junk.fa")
    open(junk, "wb").write(b"# comment line\n" + data)
    a, b = parse(junk, False), parse(junk, True)
    assert a[0] == 0 and a[:4] == b[:4] and a[1] == n
