"""Every pair kernel that can run is exercised against the oracle: the library picks the matrix-core kernels by default, so the
VALU tile kernel (fallback for `TRACS_MFMA=0`, for general alignments whose sparse lists are unavailable or too dense) and the
alternative tile shapes only run under their switches.  The switches are read once per process: each variant is a child process
that checks plain and thresholded passes of both encodings against the oracle and prints which kernel it used."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from tracs_amd import device as dev, synth
used = set()
classes = set()
for n, L, p_partial in ((300, 20000, 0.0), (300, 20000, 0.004), (131, 130000, 0.0), (131, 130000, 0.002), (70, 777, 0.02)):
    seqs = synth.alignment(n, L, seed=n + L, mu_lineage=2e-3, mu_sample=3e-4, n_lineages=5, p_n=0.02, p_partial=p_partial, p_other=0.001)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    er, ec, ed, enn = O.pairsnp_arrays(seqs, n_threads=8)
    ri, ci = er.astype(np.int64), ec.astype(np.int64)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda"); nn = torch.zeros_like(d)
    dev.pairsnp_dense(aln, d, nn)
    used.add((aln.encoding, aln.kernel))
    classes.add(aln.site_classes is not None)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32)), ("d", n, L, p_partial)
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32)), ("nn", n, L, p_partial)
    thr = int(np.percentile(ed, 30))
    d.zero_(); nn.zero_()
    dev.pairsnp_dense(aln, d, nn, dist_threshold=thr)
    keep = ed <= thr
    dh = d.cpu().numpy()
    assert np.array_equal(dh[ri[keep], ci[keep]], ed[keep].astype(np.int32)), ("thr d", n, L, p_partial)
    assert np.array_equal(nn.cpu().numpy()[ri[keep], ci[keep]], enn[keep].astype(np.int32)), ("thr nn", n, L, p_partial)
    far = dh[ri[~keep], ci[~keep]].astype(np.int64)
    assert ((far > thr) | (far < 0)).all()
    # a row panel / column block of its own (two-file geometry)
    d.zero_()
    dev.pairsnp_dense(aln, d, None, row_begin=0, row_end=n // 3, col_begin=n // 3)
    sel = (ri < n // 3) & (ci >= n // 3)
    assert np.array_equal(d.cpu().numpy()[ri[sel], ci[sel]], ed[sel].astype(np.int32))
    aln.close()
print("USED", sorted(used))
print("CLASSES", sorted(classes))
'''

VARIANTS = [
    ({}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_MFMA": "0"}, {("consensus", "valu"), ("general", "valu")}),
    ({"TRACS_GENERAL_MFMA": "0"}, {("consensus", "mfma"), ("general", "valu")}),
    ({"TRACS_FORCE_GENERAL": "1", "TRACS_GENERAL_MFMA": "1"}, {("general", "mfma-general")}),
    ({"TRACS_MFMA_TILE": "3x2"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_TILE_VARIANT": "2"}, {("consensus", "valu"), ("general", "valu")}),
    ({"TRACS_TILE_VARIANT": "3", "TRACS_KSPLIT": "3"}, {("consensus", "valu"), ("general", "valu")}),
    ({"TRACS_KSPLIT": "4", "TRACS_SUPERTILE": "2x2"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_SITE_CLASSES": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_SITE_CLASSES": "1", "TRACS_KSPLIT": "3"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_SITE_CLASSES": "1", "TRACS_MFMA_TILE": "2x2w4x2"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_MINORITY": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_COUNT_TILE": "2x2", "TRACS_KSPLIT": "2"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_COUNT_TILE": "4x2", "TRACS_MINORITY": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    # lists refused (cap of 10 entries): consensus alignments keep their classes without minority lists, general ones the VALU kernel
    ({"TRACS_LIST_CAP": "10"}, {("consensus", "mfma"), ("general", "valu")}),
    # N co-occurrences from lists (nn_rows_kernel) for every site that has two N samples, whatever the cost model says at this size;
    # never; with the threshold at a few N samples (both sources at once)
    ({"TRACS_NN_LIST_K": "1"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_NN_LIST_K": "1", "TRACS_MINORITY": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_NN_LIST_K": "1e-4", "TRACS_KSPLIT": "2"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_NN_LISTS": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    # the counting pass's source: the stored N plane in place (every site) / the counted sites' N plane re-packed
    ({"TRACS_COUNT_IN_PLACE": "1"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_COUNT_IN_PLACE": "1", "TRACS_KSPLIT": "3"}, {("consensus", "mfma"), ("general", "mfma-general")}),
    ({"TRACS_COUNT_IN_PLACE": "0"}, {("consensus", "mfma"), ("general", "mfma-general")}),
]


@pytest.mark.parametrize("env,expect", VARIANTS, ids=lambda v: "+".join("%s=%s" % kv for kv in v.items()) if isinstance(v, dict) and v else None)
def test_kernel_variant_against_oracle(hiplib, oracle, env, expect):
    full = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=full, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("USED")][-1]
    used = set(eval(line[5:]))
    assert used == expect, (used, expect)
    # site classes (csrc/site_classes.hip) only ever run with the matrix-core kernels; forced on / off by their switch
    classes = set(eval([ln for ln in out.stdout.splitlines() if ln.startswith("CLASSES")][-1][8:]))
    if env.get("TRACS_SITE_CLASSES") == "0" or all(k == "valu" for _, k in expect):
        assert classes == {False}, classes
    elif env.get("TRACS_SITE_CLASSES") == "1":
        assert classes == {True}, classes
    else:
        assert True in classes, classes
