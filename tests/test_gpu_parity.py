"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs.  Bit-exact for integer work; 1e-6 relative (BASELINE.json north_star) for floating point,
with far tighter observed agreement asserted where the arithmetic is the same."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

RTOL = 1e-6          # north_star tolerance for log-likelihoods / E(K)


@pytest.fixture(scope="module")
def api(hiplib):
    import torch  # noqa: F401  (one HIP runtime)
    from tracs_amd import api
    return api


def _write(tmp_path, seqs, name="a.fa", **kw):
    from tracs_amd import synth
    p = os.path.join(str(tmp_path), name)
    synth.write_fasta(p, seqs, **kw)
    return p


@pytest.mark.parametrize("n,L", [(2, 1), (5, 37), (17, 128), (64, 129), (65, 1000), (130, 4097), (300, 20000)])
def test_pairsnp_matches_oracle(api, oracle, tmp_path, n, L):
    from tracs_amd import synth
    seqs = synth.alignment(n, L, seed=100 + n, mu_lineage=0.02, mu_sample=0.01, p_n=0.03, p_partial=0.02,
                           p_lower=0.05, p_other=0.02)
    fa = _write(tmp_path, seqs, width=60)
    r, c, d, names, filt, nn = api.pairsnp_arrays([fa], 1, 2147483647, False)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs)
    assert names == ["s%d" % i for i in range(n)]
    assert np.array_equal(r, er) and np.array_equal(c, ec)
    assert np.array_equal(d, ed)
    assert np.array_equal(nn, enn)
    assert np.array_equal(filt, np.zeros(len(d), np.uint64))      # `len` zeros when filter is off


def test_pairsnp_threshold_and_order(api, oracle, tmp_path):
    from tracs_amd import synth
    seqs = synth.alignment(200, 3000, seed=5, mu_lineage=0.01, mu_sample=0.002, p_n=0.02)
    fa = _write(tmp_path, seqs, gz=True, name="a.fa.gz")
    for dist in (-1, 0, 3, 20, 60):
        r, c, d, _, _, nn = api.pairsnp_arrays([fa], 4, dist, False)
        er, ec, ed, enn = oracle.pairsnp_arrays(seqs, dist=dist)
        assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(d, ed) and np.array_equal(nn, enn)
        assert (d.astype(np.int64) <= dist).all()


def test_pairsnp_two_files(api, oracle, tmp_path):
    from tracs_amd import synth
    seqs = synth.alignment(90, 1500, seed=9, mu_lineage=0.02, mu_sample=0.01, p_n=0.05, p_partial=0.02)
    fa = _write(tmp_path, seqs[:37], name="q.fa", names=["q%d" % i for i in range(37)])
    fb = _write(tmp_path, seqs[37:], name="db.fa", names=["db%d" % i for i in range(53)])
    r, c, d, names, _, nn = api.pairsnp_arrays([fa, fb], 1, 40, False)
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs, n0=37, dist=40)
    assert len(names) == 90 and names[0] == "q0" and names[37] == "db0"
    assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(d, ed) and np.array_equal(nn, enn)
    assert r.max() < 37 and c.min() >= 37


def test_pairsnp_list_api_and_errors(api, tmp_path):
    from tracs_amd import synth
    seqs = synth.alignment(4, 50, seed=1)
    fa = _write(tmp_path, seqs)
    out = api.pairsnp(fasta=[fa], n_threads=1, dist=10, filter=False)
    assert isinstance(out, tuple) and len(out) == 6 and all(isinstance(x, list) for x in out)
    assert all(isinstance(v, int) for v in out[0] + out[1] + out[2] + out[4] + out[5])
    with pytest.raises(RuntimeError, match="Invalid number of fasta files!"):
        api.pairsnp(fasta=[fa, fa, fa], n_threads=1, dist=10, filter=False)
    ragged = os.path.join(str(tmp_path), "ragged.fa")
    with open(ragged, "w") as fh:
        fh.write(">a\nACGT\n>b\nACG\n")
    with pytest.raises(RuntimeError, match="variable sequence lengths"):
        api.pairsnp(fasta=[ragged], n_threads=1, dist=10, filter=False)
    empty = os.path.join(str(tmp_path), "empty.fa")
    open(empty, "w").close()
    out = api.pairsnp(fasta=[empty], n_threads=1, dist=10, filter=False)
    assert out == ([], [], [], [], [], [])
    one = os.path.join(str(tmp_path), "one.fa")
    with open(one, "w") as fh:
        fh.write(">only\nACGTN\n")
    assert api.pairsnp(fasta=[one], n_threads=1, dist=10, filter=False) == ([], [], [], ["only"], [], [])


def test_trans_dist_matches_oracle(api, oracle):
    from ek_parity import check_trans_dist
    rng = np.random.default_rng(11)
    total = {}
    for lamb, beta in ((1e-3 * 29903, 73.0), (5.3, 6.0), (3.0, 52.0)):
        N = rng.integers(0, 80, 700).astype(np.int32)     # (the oracle is serial like the reference: this test is CPU time)
        days = rng.integers(0, 500, 700)
        days[:50] = 0                                     # delta == 0 branch
        delta = days.astype(np.float64) * 86400.0 / 31556952.0
        p0, ek = api.trans_dist_arrays(N, delta, lamb, beta, 0.01)
        counts = check_trans_dist(oracle, N, delta, lamb, beta, 0.01, p0, ek)
        for k, v in counts.items():
            total[k] = total.get(k, 0) + v
    print("E(K) keys by conditioning:", total)
    assert total.get("well", 0) > 450 and total.get("saturated", 0) > 4


def test_trans_dist_reference_known_answers(api):
    # /root/reference/tests/test_trans_distance.py:29-42 (1-day gap, SNP 0 and 2, defaults)
    dd = [86400.0 / 31556952.0] * 2
    p0, ek = api.trans_dist([0, 2], dd, 1e-3 * 29903, 73.0, 0.01)
    assert abs(dd[0] - 0.002737907006988508) < 1e-15
    assert abs(np.exp(p0[0]) - 0.23794988406662973) < 1e-6 and abs(np.exp(p0[1]) - 0.024467137572328577) < 1e-6
    assert abs(ek[0] - 2.6335200453700187) < 1e-6 and abs(ek[1] - 7.315670110063259) < 1e-6
    assert isinstance(p0, list) and isinstance(ek, list)


def test_lprob_k_given_N_reference_known_answer(api):
    # /root/reference/tests/test_llk.py:21-29 (Sage symbolic integral)
    from scipy.special import gammaln
    lp, lhs = api.lprob_k_given_N(7, 4, 0.16963, 3, 52, gammaln(range(20)))
    assert abs(lp + 17.9565184209608) < 1e-6
    assert abs(lhs - 12.0861694243766) < 1e-6


def test_lprob_k_given_N_matches_oracle(api, oracle):
    from scipy.special import gammaln
    lg = gammaln(np.arange(400))
    rng = np.random.default_rng(3)
    for _ in range(40):
        N, k = int(rng.integers(0, 150)), int(rng.integers(0, 150))
        delta = float(rng.choice([0.0, 0.0027, 0.1, 0.8, 2.5]))
        a = api.lprob_k_given_N(N, k, delta, 5.3, 6.0, lg)
        b = oracle.lprob_k_given_N(N, k, delta, 5.3, 6.0, lg)
        assert np.allclose(a, b, rtol=1e-9, atol=0)


@pytest.mark.parametrize("keep", [False, True])
def test_calculate_posteriors_matches_oracle(api, oracle, keep):
    from tracs_amd import synth
    counts = synth.allele_counts(50000, seed=4, depth=25, p_two=0.05).astype(np.float64)
    counts[:50] = 0                                       # zero coverage rows
    counts[50:100] = 7                                    # four-way ties
    counts[100:150, 1] = counts[100:150, 0]               # two-way ties
    alphas = [0.889048781117318, 20.8156311152126, 0.1, 4.38181182238621]
    for thr in (0.0, 0.01, 0.05, 0.3):
        got = api.calculate_posteriors(counts, alphas, keep, thr)
        exp = oracle.calculate_posteriors(counts, alphas, keep, thr)
        assert got.shape == exp.shape and got.dtype == np.float64
        assert np.array_equal(got, exp)                   # one add + one IEEE divide per cell: bit-exact


def test_connected_components_matches_scipy_order(api, oracle):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(8)
    for n, m in ((1, 0), (10, 0), (50, 30), (2000, 1500), (20000, 30000)):
        I = rng.integers(0, n, m).astype(np.int32)
        J = rng.integers(0, n, m).astype(np.int32)
        nc, lab = api.connected_components(n, I, J)
        G = csr_matrix((np.ones(m), (I, J)), shape=(n, n))
        enc, elab = connected_components(csgraph=G, directed=False, return_labels=True)
        assert nc == enc and np.array_equal(lab, elab)
        assert np.array_equal(lab, oracle.connected_components(n, I, J))


GRID_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from tracs_amd import device as dev, synth
n = 1500
g = torch.Generator(device="cuda"); g.manual_seed(5)
d = torch.randint(0, 400, (n, n), generator=g, device="cuda", dtype=torch.int32)
d[torch.rand((n, n), generator=g, device="cuda") < 0.3] = 7            # many repeated keys
_, days_np = synth.dates(n, seed=11)
days = torch.from_numpy(days_np).cuda()
out = []
for ranges, thr, col in (([(0, n)], 2147483647, 0), ([(100, 333), (901, 1500)], 250, 64)):
    p = torch.full((n, n), -7.0, dtype=torch.float64, device="cuda"); e = torch.full((n, n), -7.0, dtype=torch.float64, device="cuda")
    dev.trans_dist_dense_ranges(d, n, days, 29.903, 73.0, 0.01, p, e, ranges, exp_p0=True, dist_threshold=thr, col_begin=col)
    out += [p.cpu().numpy(), e.cpu().numpy()]
np.savez(sys.argv[1], *out)
'''


def test_trans_dist_dense_grid_route_equals_hash_route(hiplib, tmp_path):
    """Dense blocks mark their distinct (N, day gap) keys in a grid (csrc/transcluster.hip: tc_mark_kernel) instead of hashing
    them; TRACS_TC_GRID=0 keeps the hash route.  Same keys, same evaluation: every cell bit-equal, untouched cells untouched --
    whole matrix, and two row panels with a distance threshold and a column bound."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, env in (("grid", {}), ("hash", {"TRACS_TC_GRID": "0"})):
        f = os.path.join(str(tmp_path), name + ".npz")
        out = subprocess.run([sys.executable, "-c", GRID_CHILD % {"root": root}, f], capture_output=True, text=True,
                             env=dict(os.environ, **env), timeout=900, cwd=root)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        got[name] = np.load(f)
    for k in got["grid"].files:
        a, b = got["grid"][k], got["hash"][k]
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), k
    assert (got["grid"]["arr_0"] != -7.0).sum() == 1500 * 1499 // 2
