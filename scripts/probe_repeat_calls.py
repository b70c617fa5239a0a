"""Probe: the same handle called again and again (mark_packed before every call, as bench.py's timed steps do): wall clock per call.
usage: [PARTIAL=0.005] probe_repeat_calls.py [samples] [sites] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import device as dev, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = torch.device("cuda", 0)
kw = dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=float(os.environ.get("PARTIAL", "0")))
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=20241024, **kw)
dm = torch.zeros((n, n), dtype=torch.int32, device=d)
nm = torch.zeros((n, n), dtype=torch.int32, device=d)
for k in range(calls):
    aln.mark_packed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.pairsnp_dense(aln, dm, nm)
    torch.cuda.synchronize()
    print("call %d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
aln.close()
