"""bench.py's launch contract on CPU (no GPU is touched): `python bench.py --gpus N` starts its own N ranks as a child process and
relays their line and exit code; under a launcher WORLD_SIZE must equal --gpus; --gpus 1 stays one process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK")):
    env = dict(os.environ)
    for k in drop:
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)


def test_gpus_n_starts_n_ranks_without_a_launcher():
    for n in (2, 3):
        out = _run(["--gpus", str(n), "--launch-check"])
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                                  # rank 0's line only
        j = json.loads(lines[0])
        assert j["n_gpus"] == n and j["world_size_env"] == n and j["gpus_arg"] == n


def test_gpus_1_is_one_process():
    out = _run(["--gpus", "1", "--launch-check"])
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


LAUNCHER = {"LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}     # what torch.distributed.run exports beside RANK / WORLD_SIZE


def test_world_size_must_equal_gpus():
    out = _run(["--gpus", "8", "--launch-check"], dict(LAUNCHER, RANK="0", WORLD_SIZE="1"))
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    out = _run(["--gpus", "2", "--launch-check"], dict(LAUNCHER, RANK="0", WORLD_SIZE="3"))
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_ambient_rank_variables_are_not_a_launcher():
    """RANK / WORLD_SIZE without a launcher's rendezvous (a SLURM step, somebody else's environment) do not make this process a rank"""
    out = _run(["--gpus", "1", "--launch-check"], {"RANK": "2", "WORLD_SIZE": "4"})
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_failing_rank_fails_the_command():
    """No GPU in this container: the ranks of a real run cannot start -- the command must say so and exit non-zero, not print a line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    out = _run(["--gpus", "2", "--samples", "100", "--sites", "1000", "--no-extras", "--no-cpu-baseline"])
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
