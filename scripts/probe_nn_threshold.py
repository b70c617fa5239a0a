"""Where the N lists stop paying against the matrix cores: 10 000 x 5 Mbp with N at rate P_N (cN ~ P_N x n per site), the dense
call's kernels with the lists (TRACS_NN_LIST_K=1: always) and without (TRACS_NN_LISTS=0); one process per setting (the switches are
read once).  usage: TRACS_NN_LIST_K=... python scripts/probe_nn_threshold.py <p_n>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from tracs_amd import _lib, device as dev, synth  # noqa: E402

p_n = float(sys.argv[1])
n, L = int(os.environ.get("N", "10000")), int(os.environ.get("L", "5000000"))
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=7, mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=p_n, p_partial=0.0)
dm = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nm = torch.zeros_like(dm)
lib = _lib.load()
lib.tracs_debug_pair_timing(1)
lib.tracs_debug_pack_timing(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.pairsnp_dense(aln, dm, nm)
torch.cuda.synchronize()
first = time.perf_counter() - t0
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    dev.pairsnp_dense(aln, dm, nm)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
out = (C.c_float * 4)()
lib.tracs_debug_last_pair_ms(out)
print("p_n %.4f  K=%s lists=%s: first %.1f ms, pass %.2f ms (pair %.2f fixup %.2f count %.2f nn lists %.2f)  classes %s source %s  nn checksum %d"
      % (p_n, os.environ.get("TRACS_NN_LIST_K", "-"), os.environ.get("TRACS_NN_LISTS", "1"), first * 1e3, best * 1e3, out[0], out[1], out[2], out[3],
         aln.site_classes, aln.count_source, int(nm.sum().item())))
