"""The post-pileup stage of `tracs align` on the device: allele counts -> find_dirichlet_priors ->
calculate_posteriors -> coverage rules -> IUPAC code (tracs/align.py:536-622), and packing the codes straight into the
alignment planes without the FASTA round trip.  The numpy restatement below follows the reference lines cited and uses
the oracle for the posterior filter; the GPU chain must reproduce its sequence letter for letter, and pairsnp on the
packed codes must equal pairsnp on the FASTA made from those letters."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# np.packbits(bitorder="little") index -> letter (tracs/align.py:285-323)
_LUT = np.frombuffer(b"XACMGRSVTWYHKDBN", dtype=np.uint8)


def _reference_sequence(oracle, counts, alphas, keep, thr, min_cov, band):
    post = oracle.calculate_posteriors(counts.astype(np.float64), alphas, keep, thr)      # align.py:575-577
    rs = counts.sum(1)
    if band is not None:                                                                   # :599-612
        post[(rs <= band[1]) & (rs >= band[0])] = 1
    post[rs < min_cov] = 1                                                                 # :613
    idx = np.packbits(post > 0, axis=1, bitorder="little").flatten()                      # :616-622
    return _LUT[idx]


def test_counts_to_codes_to_planes(hiplib, oracle, tmp_path):
    import torch
    from tracs_amd import api, synth
    from tracs_amd import device as dev
    L, n = 30011, 6
    alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
    seqs = []
    aln = dev.Alignment(n, L)
    for s in range(n):
        counts = synth.allele_counts(L, seed=100 + s, depth=12 + 6 * s, p_two=0.03)
        counts[5 * s:5 * s + 40] = 0                                                       # uncovered stretch
        keep = bool(s & 1)
        band = (2.0, 4.0) if s == 3 else None
        thr = 0.2 + 0.05 * (s % 3)
        ref = _reference_sequence(oracle, counts, alphas, keep, thr, 3, band)
        codes = dev.posterior_codes_device(torch.from_numpy(counts.view(np.int16)).cuda(), alphas, keep, thr, min_cov=3,
                                           cov_band=band)
        got = dev.codes_to_iupac_device(codes, L).cpu().numpy()
        assert np.array_equal(got, ref), (s, np.where(got != ref)[0][:5])
        assert (ref == ord("N")).any() and (ref == ord("A")).any()
        aln.pack_codes(codes, s)
        seqs.append(ref)
    seqs = np.array(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    fa = os.path.join(str(tmp_path), "codes.fa")
    synth.write_fasta(fa, seqs)
    r, c, ed, names, _, enn = api.pairsnp_arrays([fa], 1, 2147483647, False)
    ri, ci = r.astype(np.int64), c.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    orr, occ, od, onn = oracle.pairsnp_arrays(seqs)
    assert np.array_equal(ed, od) and np.array_equal(enn, onn)
