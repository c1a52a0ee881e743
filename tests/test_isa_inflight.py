"""The fused MLP / wgrad kernels issue loads as inline asm and wait for them with hand-counted s_waitcnt many
instructions later; the compiler must never touch a register whose load may still be outstanding.
scripts/check_inflight.py verifies that on the final ISA (hipcc cross-compiles without a GPU).  The eleven sources are
compiled to assembly once, side by side (a minute or two instead of five), then every file is checked on its own."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ["mlp_fwd.hip", "mlp_bwd.hip", "wgrad.hip", "mlp_fwd_bf16.hip", "mlp_fwd_bf16_t16.hip", "mlp_bwd_bf16.hip", "wgrad_bf16.hip",
           "mlp_fwd_f16x3.hip", "mlp_fwd_f16x3_t16.hip", "mlp_bwd_f16x3.hip", "wgrad_f16x3.hip"]


def _compile(src, out_dir):
    asm = os.path.join(out_dir, src + ".s")
    # flags must match nerf_amd/csrc/Makefile
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
           "-o", asm, os.path.join(ROOT, "nerf_amd", "csrc", src)]
    if "f16x3" in src:
        cmd.insert(6, "-fno-slp-vectorize")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    return src, asm, r.returncode, r.stderr[-2000:]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    if not os.path.exists(HIPCC) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out_dir = str(tmp_path_factory.mktemp("isa"))
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        done = list(pool.map(lambda s: _compile(s, out_dir), SOURCES))
    return {src: (asm, rc, err) for src, asm, rc, err in done}


@pytest.mark.parametrize("src", SOURCES)
def test_no_instruction_touches_inflight_registers(src, assembly):
    asm, rc, err = assembly[src]
    assert rc == 0, err
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_inflight.py"), asm],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "in-flight register violations" in r.stdout
