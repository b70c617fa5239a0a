"""find_dirichlet_priors (tracs/dirichlet_multinomial.py:9-73): oracle vs the reference's outputs (golden) and the
R MGLM known answer of the reference's own test; GPU vs oracle/golden."""
import json
import os

import numpy as np
import pytest


@pytest.fixture()
def gold(golden_dir):
    with open(os.path.join(golden_dir, "python_reference_golden.json")) as fh:
        return json.load(fh)["find_dirichlet_priors"]


def _counts4(seed):
    from tracs_amd import synth
    return synth.allele_counts(3000, seed=seed, depth=25, p_two=0.06).astype(float)


def test_oracle_vs_reference_outputs(oracle, gold):
    c3 = np.array(gold["counts3"], float)
    fp = oracle.find_dirichlet_priors(c3, tol=1e-10, method="FP")
    loo = oracle.find_dirichlet_priors(c3, tol=1e-10, method="LOO")
    assert np.allclose(fp, gold["fp"], rtol=1e-10) and np.allclose(loo, gold["loo"], rtol=1e-10)
    # the reference's own assertion (tests/test_dirichlet_multinomial.py:10-18): one-sided, vs R MGLM
    assert np.max(fp - np.array(gold["r_mglm"])) < 1e-3 and np.max(loo - np.array(gold["r_mglm"])) < 1e-3
    assert np.allclose(fp, gold["r_mglm"], rtol=1e-3)
    c4 = _counts4(gold["counts4_seed"])
    assert np.allclose(oracle.find_dirichlet_priors(c4, method="FPI", error_filt_threshold=0.01), gold["fp4_filt0.01"], rtol=1e-10)
    assert np.allclose(oracle.find_dirichlet_priors(c4[:40], method="FPI"), gold["fp4_first40"], rtol=1e-10, atol=1e-15)
    assert np.array_equal(oracle.find_dirichlet_priors(c4[:12] * np.array([0, 0, 0, 1.0])), np.array([0, 0, 0, 1.0]))


@pytest.mark.gpu
def test_gpu_vs_reference_outputs(hiplib, oracle, gold, capsys):
    import torch  # noqa: F401
    from tracs_amd.dirichlet_multinomial import find_dirichlet_priors
    c3 = np.array(gold["counts3"], float)
    fp = find_dirichlet_priors(c3, tol=1e-10, method="FP")
    loo = find_dirichlet_priors(c3, tol=1e-10, method="LOO")
    assert "Calculated alphas" in capsys.readouterr().out
    assert np.allclose(fp, gold["fp"], rtol=1e-8) and np.allclose(loo, gold["loo"], rtol=1e-8)
    assert np.max(fp - np.array(gold["r_mglm"])) < 1e-3 and np.max(loo - np.array(gold["r_mglm"])) < 1e-3
    c4 = _counts4(gold["counts4_seed"])
    assert np.allclose(find_dirichlet_priors(c4, method="FPI", error_filt_threshold=0.01), gold["fp4_filt0.01"], rtol=1e-8)
    assert np.allclose(find_dirichlet_priors(c4[:40], method="FPI"), gold["fp4_first40"], rtol=1e-8, atol=1e-15)
    assert np.array_equal(find_dirichlet_priors(c4[:12] * np.array([0, 0, 0, 1.0])), np.array([0, 0, 0, 1.0]))
    # larger table, both methods, default tolerance: vs the oracle
    from tracs_amd import synth
    big = synth.allele_counts(400000, seed=77, depth=30, p_two=0.03).astype(float)
    for method in ("FPI", "LOO"):
        for filt in (None, 0.02):
            a = find_dirichlet_priors(big, method=method, error_filt_threshold=filt)
            b = oracle.find_dirichlet_priors(big, method=method, error_filt_threshold=filt)
            assert np.allclose(a, b, rtol=1e-7), (method, filt, a, b)
