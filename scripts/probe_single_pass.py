"""Probe: cold / warm single pass (packed planes resident -> d, nn, P, E(K)) with the per-pack stage trace.
usage: probe_single_pass.py [samples] [sites]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import _lib, device as dev, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
d = torch.device("cuda", 0)
seed = 20241022 + 2
kw = dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=float(os.environ.get("PARTIAL", "0")))
_, days_np = synth.dates(n, seed=seed)
days = torch.from_numpy(days_np).to(d)
dm = torch.zeros((n, n), dtype=torch.int32, device=d)
nm = torch.zeros((n, n), dtype=torch.int32, device=d)
pm = torch.zeros((n, n), dtype=torch.float64, device=d)
em = torch.zeros((n, n), dtype=torch.float64, device=d)


def one_pass(tag):
    aln = dev.Alignment(n, L)
    synth.pack_synthetic_device(aln, seed=seed, **kw)
    _lib.load().tracs_debug_pack_timing(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.pairsnp_dense(aln, dm, nm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev.trans_dist_dense_ranges(dm, n, days, 29.903, 73.0, 0.01, pm, em, [(0, n)], exp_p0=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: pairsnp %.1f ms, transcluster %.1f ms, total %.1f ms  checksum %d" %
          (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, int(dm.sum().item())), flush=True)
    print("   stages:", ", ".join("%s %.2f" % x for x in dev.pack_stages()), " classes", aln.site_classes, aln.count_source, flush=True)
    t0 = time.perf_counter()
    dev.pairsnp_dense(aln, dm, nm)
    dev.trans_dist_dense_ranges(dm, n, days, 29.903, 73.0, 0.01, pm, em, [(0, n)], exp_p0=True)
    torch.cuda.synchronize()
    print("%s: repeat pass %.1f ms" % (tag, (time.perf_counter() - t0) * 1e3), flush=True)
    import ctypes as C
    out = (C.c_float * 4)()
    lib = _lib.load(); lib.tracs_debug_pair_timing(1)
    dev.pairsnp_dense(aln, dm, nm); torch.cuda.synchronize()
    if lib.tracs_debug_last_pair_ms(out) == 0:
        print("   kernels: pair %.2f  fixup %.2f  count %.2f  nn lists %.2f ms   checksum nn %d" % (out[0], out[1], out[2], out[3], int(nm.sum().item())), flush=True)
    aln.close()


one_pass("cold")
one_pass("warm")
if len(sys.argv) <= 3 and not os.environ.get("PARTIAL"):
    one_pass("warm2")
