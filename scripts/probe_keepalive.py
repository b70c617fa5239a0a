"""Probe: second handle packed while the first is alive (bench.py's warm single pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TRACS_CLASSES_TRACE"] = "1"
import torch
from tracs_amd import _lib, device as dev, synth
n, L = 10000, 5000000
d = torch.device("cuda", 0)
seed = 20241024
kw = dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=0.0)
dm = torch.zeros((n, n), dtype=torch.int32, device=d); nm = torch.zeros((n, n), dtype=torch.int32, device=d)
keep = []
for tag in ("first", "second(first alive)", "third(both alive)"):
    aln = dev.Alignment(n, L)
    synth.pack_synthetic_device(aln, seed=seed, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.pairsnp_dense(aln, dm, nm)
    torch.cuda.synchronize()
    print("%s: pairsnp %.1f ms" % (tag, (time.perf_counter() - t0) * 1e3), flush=True)
    keep.append(aln)
