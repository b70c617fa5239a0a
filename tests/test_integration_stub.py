"""INTEGRATION.md section 2 shows the ~40-line ctypes binding a TRACS maintainer would add.  This test executes that
very block against the built library, so the document cannot drift from the ABI."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace(monkeypatch):
    from tracs_amd import _lib
    _lib.load()                                              # shares the HIP runtime with torch first (see _lib.py)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# tracs/_hip\.py.*?)```", text, re.S).group(1)
    monkeypatch.setenv("TRACS_HIP_LIB", _lib.LIB_PATH)
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    return ns


def test_stub_binds_every_symbol_it_names(hiplib, monkeypatch):
    ns = _stub_namespace(monkeypatch)                        # CDLL + restype/argtypes lines run without a GPU
    assert callable(ns["pairsnp"]) and callable(ns["trans_dist"]) and callable(ns["calculate_posteriors"])


@pytest.mark.gpu
def test_stub_results_equal_the_package(hiplib, monkeypatch, tmp_path):
    from tracs_amd import api, synth
    ns = _stub_namespace(monkeypatch)
    seqs = synth.alignment(12, 5003, seed=3, p_n=0.02, p_partial=0.01)
    fa = os.path.join(str(tmp_path), "a.fa")
    synth.write_fasta(fa, seqs)
    assert ns["pairsnp"]([fa], 1, 2147483647, False) == api.pairsnp([fa], 1, 2147483647, False)
    n = np.array([0, 3, 17, 40], dtype=np.int64)
    d = np.array([0.0, 0.01, 0.5, 1.25])
    assert ns["trans_dist"](n, d, 5.3, 6.0, 0.01) == api.trans_dist(n, d, 5.3, 6.0, 0.01)
    counts = np.random.default_rng(1).integers(0, 30, (1000, 4)).astype(float)
    al = [20.8, 4.4, 0.9, 0.1]
    assert np.array_equal(ns["calculate_posteriors"](counts, al, False, 0.01), api.calculate_posteriors(counts, al, False, 0.01))
