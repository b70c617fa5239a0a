"""ASan + UBSan, then TSan, over the host-only sources of libtracs_hip.so (FASTA readers, pileup parser, CSV writers/reader, combine):
scripts/sanitize_host.sh builds them with g++ -fsanitize=address,undefined and drives them with adversarial inputs.
(GPU-side sanitizers are not available on the pool; the kernels are covered by the parity tests.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_sources_clean_under_asan_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_host.sh")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("all host paths clean") == 2          # ASan+UBSan pass, then TSan pass
