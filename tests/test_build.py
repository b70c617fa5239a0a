"""build() really compiles: a from-scratch build into a scratch directory produces a gfx950 shared library that exports
the whole C ABI (the in-tree library the other tests have loaded is left alone), and the driver hook forces a rebuild."""
import ctypes as C
import os
import subprocess
import time


def test_forced_build_compiles_every_source(tmp_path):
    from tracs_amd import _lib
    from tracs_amd import build as b
    t0 = time.time()
    lib = b.build(force=True, libdir=str(tmp_path))
    assert os.path.dirname(lib) == str(tmp_path) and os.path.getmtime(lib) >= t0 - 1.0
    objs = sorted(f for f in os.listdir(tmp_path) if f.endswith(".o"))
    assert objs == sorted(os.path.splitext(s)[0] + ".o" for s in b.SOURCES)          # one object per source, all fresh
    assert all(os.path.getmtime(os.path.join(tmp_path, o)) >= t0 - 1.0 for o in objs)
    # gfx950 code object inside, and every declared symbol exported
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", lib], capture_output=True, text=True).stdout
    assert "gfx950" in out
    L = C.CDLL(lib)
    for name in _lib.SYMBOLS:
        assert hasattr(L, name), name


def test_in_tree_rebuild_moves_the_mtime():
    """The hook the driver calls (__graft_entry__.build) rebuilds the in-tree library instead of trusting the file that
    travelled with the snapshot; run in a child process so this process's mapping of the old file is not disturbed."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tracs_amd", "lib", "libtracs_hip.so")
    before = os.path.getmtime(lib) if os.path.exists(lib) else 0.0
    time.sleep(1.1)
    rc = subprocess.run(["python", "-c", "import __graft_entry__ as g; g.build()"], cwd=root, capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr[-2000:]
    assert os.path.getmtime(lib) > before
