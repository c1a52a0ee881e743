"""The plain-C client of the C ABI (tests/c_abi_driver.c) on the GPU: create -> set_params -> render -> render_u8 ->
train_backward -> optimizer_step -> get_params -> destroy from a C11 program with no Python or torch in the process,
compared with the ctypes path on the same inputs.  Render outputs are bit-identical (deterministic kernels); the
gradient is combined with fp32 atomics whose arrival order varies, so gradients / updated parameters agree to
rounding (the bar of test_adam_step_matches_oracle's resume check)."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,dtype", [("small_dir", "f32"), ("small_sh", "f32"), ("small_dir", "f16x3"), ("small_dir", "bf16")])
def test_c_driver_equals_ctypes_path(tmp_path, name, dtype):
    import gpu_util as G
    from nerf_amd import NerfConfig
    drv = os.path.join(ROOT, "tests", "bin", "c_abi_driver")
    assert os.path.exists(drv), "tests/bin/c_abi_driver is missing: run __graft_entry__.build()"
    c = H.CONFIGS[name]
    n = 200
    ro, rd, b, label = H.make_rays(n, 17)
    params = H.make_params(c, 4)
    cfg = NerfConfig.from_dict(dict(c, dtype=dtype))
    cfg.maxRays = 256
    d = str(tmp_path)
    open(os.path.join(d, "config.bin"), "wb").write(bytes(cfg.to_c()))
    for nm, a in (("params", params), ("rays_o", ro), ("rays_d", rd), ("bounds", b), ("label", label)):
        np.ascontiguousarray(a, np.float32).tofile(os.path.join(d, nm + ".bin"))
    r = subprocess.run([drv, "run", d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_driver ok" in r.stdout
    rd_ = lambda nm, dt=np.float32: np.fromfile(os.path.join(d, nm + ".bin"), dtype=dt)
    blk = G.make_block(dict(c, dtype=dtype), max_rays=256)
    blk.set_params(params)
    fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, debug=True)
    np.testing.assert_array_equal(rd_("rgb").reshape(n, 3), fine)
    np.testing.assert_array_equal(rd_("depth"), dbg["depth"])
    np.testing.assert_array_equal(rd_("u8", np.uint8).reshape(n, 3), blk.render_u8(ro, rd, b, rd))
    # the frame the C program rendered from its pose (whole == three row blocks was checked there, in C)
    c2w = np.array([[1, 0, 0, 0.2], [0, 0.6, -0.8, -3.2], [0, 0.8, 0.6, 2.4]], np.float32)
    np.testing.assert_array_equal(rd_("pose", np.uint8).reshape(26, 10, 3), blk.render_pose(26, 10, 22.0, c2w, 2.0, 6.0, False))
    lf, lc = blk.train_backward(ro, rd, b, rd, label)
    g = blk.get_grads()
    blk.optimizer_step()
    loss = rd_("loss")
    assert loss[0] == pytest.approx(lf, rel=1e-6) and loss[1] == pytest.approx(lc, rel=1e-6)
    np.testing.assert_allclose(rd_("grads"), g, rtol=0, atol=2e-6 * np.abs(g).max())
    np.testing.assert_allclose(rd_("params_after"), blk.get_params(), rtol=0, atol=2e-4 if dtype == "bf16" else 2e-6)
    blk.close()
