"""The two-stream API (include/tracs_hip.h "Two streams"): tracs_pairsnp_notify_distances + tracs_set_stream_policy.

transcluster only reads the distances, which are final before the compared-sites counts are; a caller may run it on a second
stream behind the event the dense call records once d is final.  The result must equal the single-stream one."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _setup(n, L, seed):
    import torch
    from tracs_amd import device as dev, synth
    d = torch.device("cuda", 0)
    seqs = synth.alignment(n, L, seed=seed, mu_lineage=0.0, mu_sample=2e-3, n_lineages=1, p_n=0.02)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    _, days_np = synth.dates(n, seed=seed)
    days = torch.from_numpy(days_np).to(d)
    return torch, dev, d, aln, days


def test_unrecorded_event_gets_a_handle(hiplib):
    import torch
    from tracs_amd import device as dev
    e = torch.cuda.Event()
    dev.notify_distances(e)                   # must not hand NULL to the library (torch creates the handle lazily)
    assert e.cuda_event != 0
    hiplib.tracs_pairsnp_notify_distances(None)      # disarm


@pytest.mark.parametrize("n,L", [(700, 40000)])
def test_transcluster_on_a_second_stream_behind_the_distances(hiplib, n, L):
    torch, dev, d, aln, days = _setup(n, L, 41)
    mats = lambda dt: [torch.zeros((n, n), dtype=dt, device=d) for _ in range(2)]
    (d1, n1), (p1, e1) = mats(torch.int32), mats(torch.float64)
    (d2, n2), (p2, e2) = mats(torch.int32), mats(torch.float64)
    # single stream
    dev.pairsnp_dense(aln, d1, n1)
    dev.trans_dist_dense_ranges(d1, n, days, 29.903, 73.0, 0.01, p1, e1, [(0, n)], exp_p0=True)
    torch.cuda.synchronize()
    # two streams: the caller orders them with the event
    side = torch.cuda.Stream(device=d)
    ready = torch.cuda.Event()
    dev.set_stream_policy(True)
    try:
        for _ in range(3):                    # repeated: the second and third calls race the previous transcluster without the waits
            torch.cuda.current_stream().wait_stream(side)
            dev.notify_distances(ready)
            dev.pairsnp_dense(aln, d2, n2)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                dev.trans_dist_dense_ranges(d2, n, days, 29.903, 73.0, 0.01, p2, e2, [(0, n)], exp_p0=True)
        torch.cuda.synchronize()
    finally:
        dev.set_stream_policy(False)
    up = torch.triu(torch.ones((n, n), dtype=torch.bool, device=d), diagonal=1)
    assert torch.equal(d1[up], d2[up]) and torch.equal(n1[up], n2[up])
    assert torch.equal(p1[up], p2[up]) and torch.equal(e1[up], e2[up])
    assert float(e1[up].max()) > 0.0
    aln.close()


def test_event_is_recorded_on_early_returns(hiplib):
    """An armed event never survives its call: an empty row range and an argument error both record (or drop) it."""
    torch, dev, d, aln, days = _setup(64, 2000, 5)
    dm = torch.zeros((64, 64), dtype=torch.int32, device=d)
    e = torch.cuda.Event()
    dev.notify_distances(e)
    dev.pairsnp_dense(aln, dm, None, row_begin=10, row_end=10)       # no rows: returns at once
    torch.cuda.synchronize()
    assert e.query()
    e2 = torch.cuda.Event()
    dev.notify_distances(e2)
    with pytest.raises(RuntimeError):
        import ctypes as C
        from tracs_amd import _lib
        _lib.check(hiplib.tracs_pairsnp_dense(aln._h, 0, 64, 0, C.c_void_p(dm.data_ptr()), None, 8, None))     # ld < n
    # a later, unrelated call must not record e2 again after we re-record it ourselves
    torch.cuda.synchronize()
    assert e2.query()
    aln.close()
