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


def test_roofline_traffic_is_null_for_a_library_built_from_other_sources(tmp_path, monkeypatch):
    """bench.py quotes counter traffic only from a PMC summary whose recorded nerfhip_version() (release
    + hash of the kernel sources) equals the loaded library's; any other summary is stale and yields null with the reason."""
    sys.path.insert(0, ROOT)
    import bench
    from nerf_amd import _lib
    mine = _lib.load().nerfhip_version().decode()
    assert " src " in mine and len(mine.rsplit(" ", 1)[1]) == 12          # a 12-hex-digit source hash rides in the version
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    hdr = "kernel,dispatches,total_ms,FETCH_SIZE\n"
    (prof / "rX_f32_pmc_fetchsize.csv").write_text(hdr + '"void nh::mlp_fwd_f32_kernel<8, true>",6,1.0,3000\n')
    (prof / "rX_f32_pmc_writesize.csv").write_text(hdr.replace("FETCH", "WRITE") + '"void nh::mlp_fwd_f32_kernel<8, true>",6,1.0,6000\n')
    (prof / "rX_f32_pmc_meta.json").write_text(json.dumps({"nerfhip_version": "nerfhip 0.3 (other) src 000000000000"}))
    r = bench.pmc_traffic("mlp_fwd", "f32")
    assert r["traffic"] is None and "other sources" in r["traffic_note"]
    (prof / "rX_f32_pmc_meta.json").write_text(json.dumps({"nerfhip_version": mine}))
    r = bench.pmc_traffic("mlp_fwd", "f32")
    assert r["traffic"] == round((2.0 * 3000 + 6000) * 1024 / 3.0)          # FETCH_SIZE doubled (gfx950), KB, 3 profiled steps
    assert bench.pmc_traffic("mlp_wgrad", "bf16")["traffic"] is None       # no summary of that dtype at all
