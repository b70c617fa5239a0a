"""Build libtracs_hip.so (gfx950) in-tree with hipcc.  `python -m tracs_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtracs_hip.so")
SOURCES = ["capi.hip", "pairsnp.hip", "pairsnp_mfma.hip", "general_sparse.hip", "site_lists.hip", "site_classes.hip", "transcluster.hip", "dmultinomial.hip", "cluster.hip", "filter.hip", "filter_lists.hip", "dirichlet.hip", "exchange.hip", "fasta.cpp", "alignio.cpp", "comm.cpp"]
HEADERS = ["common.h", "pairsnp_kernels.h", "fasta.h", "rowwriter.h", os.path.join("..", "..", "include", "tracs_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=(), libdir=None):
    """Compile every HIP source for gfx950 and link the shared library.  Cross-compiles without a GPU.
    TRACS_EXTRA_HIPCC_FLAGS adds flags (e.g. -DTRACS_MFMA_SWEEP: every matrix-core tile shape, for shape sweeps).
    libdir: build into another directory (tests/test_build.py compiles from scratch without touching the loaded library)."""
    extra_flags = list(extra_flags) + os.environ.get("TRACS_EXTRA_HIPCC_FLAGS", "").split()
    out_dir = LIBDIR if libdir is None else libdir
    lib = os.path.join(out_dir, "libtracs_hip.so")
    if libdir is None and not force and not _stale():
        return lib
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + FLAGS + list(extra_flags) + (["-x", "hip"] if src.endswith(".hip") else []) + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("libtracs_hip.so: compilation failed")
    # link next to the target and rename over it: a process that has the old file mapped keeps its own copy
    tmp = lib + ".tmp.%d" % os.getpid()
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] + objs + ["-lz", "-ldl"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("libtracs_hip.so: link failed:\n" + out.stdout + out.stderr)
    os.replace(tmp, lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
