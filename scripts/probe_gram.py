import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from tracs_amd import device as dev
from test_gpu_nw_gram import _masked
for (n, L, pp, lin) in [(200, 20000, 0.01, 2), (320, 30011, 0.005, 1), (1100, 6000, 0.004, 3), (2000, 8000, 0.005, 2), (640, 20000, 0.01, 2)]:
    seqs = _masked(n, L, seed=n + L, p_partial=pp, lineages=lin)
    aln = dev.Alignment(n, L); aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda"); nn = torch.zeros_like(d)
    dev.pairsnp_dense(aln, d, nn)
    print(n, L, pp, lin, aln.site_classes, aln.nw_gram, aln.kernel, aln.count_source)
