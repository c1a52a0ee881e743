"""bench.py end to end on the one GPU of the test box, small sizes: the contract's keys are there, the data-parallel code
path (a 1-rank RCCL communicator, --force-dp) runs, the sharded 800x800 leg produces a whole frame and the native-RCCL
comparison behind the headline agrees with the torch path.  (The N > 1 launch logic itself is covered without a GPU by
tests/test_bench_launch.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_small_run_with_the_data_parallel_path():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = "29671"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dp", "--steps", "2", "--warmup", "1",
                        "--rays", "512", "--hw", "64", "--render-frames", "1", "--no-cpu-baseline", "--no-bf16-render",
                        "--no-bf16-train"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"] == "rays/sec (train step)" and d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32"
    assert d["value"] > 0 and d["extras_error"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["config"]["parallelism"] == "dp1" and "all_reduce" in d["config"]["exchange"]
    s8 = d["render_sharded_800"]
    assert s8["rows_per_rank"] == 800 and s8["n_gpus"] == 1 and s8["value"] > 0 and s8["image_checksum"] > 0
    nat = [l for l in r.stderr.splitlines() if l.startswith("[bench native_rccl] {")]
    assert len(nat) == 1, r.stderr[-2000:]
    n = json.loads(nat[0][len("[bench native_rccl] "):])
    assert n["ok"] is True and n["n_gpus"] == 1 and n["max_abs_param_diff_vs_torch"] < 1e-3
