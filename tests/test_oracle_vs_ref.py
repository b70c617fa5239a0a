"""Live comparison of the oracle with oracle/_ref (the reference's transcluster.hpp / dmultinomial.hpp compiled
in place).  Runs in a SUBPROCESS: the reference is built with -ffast-math, and loading such a library
switches the whole process to flush-to-zero arithmetic.  Skipped where _ref is not present."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
R = O.ref_module()
assert R is not None
rng = np.random.default_rng(5)
bad = 0
for lamb, beta in ((5.3, 6.0), (29.903, 73.0), (3.0, 52.0)):
    N = rng.integers(0, 90, 400); days = rng.integers(0, 300, 400); days[:20] = 0
    delta = days * 86400.0 / 31556952.0
    p0, ek = O.trans_dist(N, delta, lamb, beta, 0.01)
    rp0, rek = R.ref_trans_dist(N.tolist(), delta.tolist(), lamb, beta, 0.01)
    assert np.allclose(p0, rp0, rtol=1e-12, atol=0)
    for i in range(400):
        cls = O.ek_conditioning(int(N[i]), float(delta[i]), lamb, beta, 0.01)[0]
        if cls == "well":
            assert abs(ek[i] - rek[i]) <= 1e-9 * abs(rek[i]), (lamb, N[i], days[i], ek[i], rek[i])
        elif abs(ek[i] - rek[i]) > 1e-9 * abs(rek[i]):
            bad += 1
counts = rng.poisson(6, (2000, 4)).astype(float); counts[:30] = 0; counts[30:60] = 4
for keep in (False, True):
    a = O.calculate_posteriors(counts, [3.0, 0.2, 9.0, 0.7], keep, 0.04)
    b = np.asarray(R.ref_calculate_posteriors(counts, [3.0, 0.2, 9.0, 0.7], keep, 0.04))
    assert np.max(np.abs(a - b)) <= 4e-16
print("OK ill/saturated disagreements:", bad)
'''


def test_oracle_matches_reference_build():
    so_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(so_dir) or not any(f.startswith("_tracs_ref") for f in os.listdir(so_dir)):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    out = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
