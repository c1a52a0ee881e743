"""bench.py's launch logic on a CPU box (runNerf.scala:95 `next.split(trainer.getDevices)`: the job owns
N devices).  `python bench.py --gpus N` started WITHOUT a launcher must reach N ranks by itself, and a world size that
differs from --gpus must end non-zero -- never a line that reports another n_gpus.  --dry-launch keeps the GPU out of it:
the ranks only rendezvous over gloo and rank 0 prints the world it saw."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus2_without_launcher_reaches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"dry_launch": True, "n_gpus": 2, "world_size_env": 2}
    assert "starting 2 ranks" in r.stderr


def test_gpus1_runs_in_process():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1
    assert "starting" not in r.stderr


def test_world_size_mismatch_is_an_error():
    """a launcher that brought ONE rank to a --gpus 2 request (the shape of round 2's silent 1-rank run)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in r.stderr
    assert "n_gpus" not in r.stdout
    # and the other way round: three ranks under a --gpus 1 request
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
