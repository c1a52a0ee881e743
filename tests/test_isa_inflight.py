"""The fused MLP / wgrad kernels issue loads as inline asm and wait for them with hand-counted s_waitcnt many
instructions later; the compiler must never touch a register whose load may still be outstanding.
scripts/check_inflight.py verifies that on the final ISA (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src", ["mlp_fwd.hip", "mlp_bwd.hip", "wgrad.hip", "mlp_fwd_bf16.hip", "mlp_bwd_bf16.hip", "wgrad_bf16.hip",
                                 "mlp_fwd_f16x3.hip", "mlp_bwd_f16x3.hip", "wgrad_f16x3.hip"])
def test_no_instruction_touches_inflight_registers(src, tmp_path):
    if not os.path.exists(HIPCC) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    asm = tmp_path / (src + ".s")
    # flags must match nerf_amd/csrc/Makefile
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
           "-o", str(asm), os.path.join(ROOT, "nerf_amd", "csrc", src)]
    if "f16x3" in src:
        cmd.insert(6, "-fno-slp-vectorize")
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_inflight.py"), str(asm)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "in-flight register violations" in r.stdout
