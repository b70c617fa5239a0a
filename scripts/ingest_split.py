import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tracs_amd import _lib, device as dev
L_=_lib.require_gpu()
n, L = 400, 5000000
rng = np.random.default_rng(3)
base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
path = "/tmp/split.fa"
with open(path, "wb") as f:
    for s in range(n):
        f.write(b">s%d\n" % s + b"\n".join(base[o:o + 80].tobytes() for o in range(0, L, 80)) + b"\n")
for rep in range(2):
    nn, LL = C.c_size_t(0), C.c_size_t(0)
    t0=time.perf_counter(); L_.tracs_debug_read_fasta(path.encode(), C.byref(nn), C.byref(LL), None); t1=time.perf_counter()
    a = dev.Alignment.from_fasta([path]); torch.cuda.synchronize(); t2=time.perf_counter()
    a.close()
    print("parse only %.3f s   from_fasta %.3f s" % (t1-t0, t2-t1))
x = torch.empty(2_000_000_000, dtype=torch.uint8)
t0=time.perf_counter(); y = x.cuda(); torch.cuda.synchronize(); print("pageable H2D 2 GB %.3f s" % (time.perf_counter()-t0))
xp = x.pin_memory()
t0=time.perf_counter(); y = xp.cuda(); torch.cuda.synchronize(); print("pinned H2D 2 GB %.3f s" % (time.perf_counter()-t0))
