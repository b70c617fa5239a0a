"""Probe: the list walk (nn_rows_kernel) with the sites cut into segments that all rows walk before the next one starts
(TRACS_NN_SEGMENT_MB: bytes of list lines per segment; 0 = one segment).  usage: probe_nn_segments.py [samples] [sites]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import _lib, device as dev, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
d = torch.device("cuda", 0)
seed = 20241022 + 2
kw = dict(mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01, p_partial=0.0)
dm = torch.zeros((n, n), dtype=torch.int32, device=d)
nm = torch.zeros((n, n), dtype=torch.int32, device=d)
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=seed, **kw)
lib = _lib.load()
lib.tracs_debug_pair_timing(1)
out = (C.c_float * 4)()
for mb in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1024,512,256,128,96,64,48,32,24".split(","))]:
    os.environ["TRACS_NN_SEGMENT_MB"] = str(mb)
    ts = []
    for _ in range(3):
        dev.pairsnp_dense(aln, dm, nm)
        torch.cuda.synchronize()
        lib.tracs_debug_last_pair_ms(out)
        ts.append(out[3])
    print("segment %5d MB: nn lists %s ms   checksum nn %d  d %d" % (mb, " ".join("%.2f" % t for t in ts), int(nm.sum().item()), int(dm.sum().item())), flush=True)
print(aln.list_stats if hasattr(aln, "list_stats") else "")
