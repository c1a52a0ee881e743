"""Edge cases at the boundary (task brief: empty and ragged inputs, maximum sizes): the largest sample counts
nerfhip_create accepts (the per-ray kernels stage one ray per wave in at most 64 KiB of LDS per workgroup) render and train
to the usual bars against the oracle, one sample more is refused loudly, and an empty batch is a no-op for render and an
error for training.  Reference shapes: NSamples / NImportance are plain Ints of nerfConfig (nerfConfig.scala:31-32), the
reference has no limit of its own short of memory."""
import numpy as np
import pytest

import gpu_util as G
import helpers as H
from nerf_amd import NerfBlock, NerfConfig, NerfHipError
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def test_limits_are_refused_with_a_message():
    for kw, word in ((dict(NSamples=1025, useHierarchical=False), "NSamples"),
                     (dict(NSamples=2, useHierarchical=False), "NSamples"),
                     (dict(NSamples=1024, NImportance=1024), "too large"),                  # compositing backward: 48 B of LDS per sample
                     (dict(NSamples=512, NImportance=852), "resampling"),                   # CDF + bins + sort buffer beyond 64 KiB
                     (dict(D=17), "D must"), (dict(W=48), "W must"), (dict(posL=11), "posL")):
        with pytest.raises(NerfHipError) as e:
            NerfBlock(NerfConfig.from_dict(dict(H.cfg(D=2, W=32, skips=[], posL=2, dirL=1, **{k: v for k, v in kw.items() if k not in ("D", "W", "posL")}),
                                                **{k: v for k, v in kw.items() if k in ("D", "W", "posL")}, maxRays=8)))
        assert word in str(e.value), (kw, str(e.value))


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_largest_flat_sample_count(dtype):
    """NSamples = 1024 (the maximum), no hierarchical pass: render and gradients against the oracle"""
    c = H.cfg(D=2, W=64, skips=[0], posL=3, dirL=1, NSamples=1024, NImportance=0, useHierarchical=False)
    n = 9
    ro, rd, b, label = H.make_rays(n, 41)
    params = H.make_params(c, 8)
    o = Oracle(c)
    blk = G.make_block(c, max_rays=16, dtype=dtype)
    blk.set_params(params)
    fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, debug=True)
    ref = o.forward(params, ro, rd, b, rd)
    np.testing.assert_array_equal(dbg["z_fine"], ref["z_fine"])
    np.testing.assert_allclose(fine, ref["rgb_fine"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(dbg["w_fine"], ref["w_fine"], rtol=2e-4, atol=2e-6)
    lf, _ = blk.train_backward(ro, rd, b, rd, label)
    r = o.train_grads(params, ro, rd, b, rd, label)
    assert lf == pytest.approx(r["loss_fine"], rel=1e-4)
    H.assert_grads_close(c, blk.get_grads(), r["grads"], whole=1e-3)
    blk.close()


def test_largest_hierarchical_sample_counts():
    """NSamples 256 + NImportance 1024: the resampling kernel's LDS image is exactly 64 KiB, 1280 samples per ray in the
    compositing kernels (the generic LDS form: more than 320).  Sampler bit for bit, render and loss to the usual bars."""
    c = H.cfg(D=2, W=32, skips=[], posL=3, dirL=1, NSamples=256, NImportance=1024)
    n = 7
    ro, rd, b, label = H.make_rays(n, 43)
    params = H.make_params(c, 9)
    o = Oracle(c)
    blk = G.make_block(c, max_rays=8)
    blk.set_params(params)
    # the sampler alone on identical (w, z): bit-exact
    w = (H.det_uniform(n * 256, 3) ** 4).reshape(n, 256).astype(np.float32)
    z = np.sort(H.det_uniform(n * 256, 4, 2.0, 6.0).reshape(n, 256), axis=1)
    cdf, ind, zs, zf = o.sample_pdf(w, z)
    g_cdf, g_ind, g_zs, g_zf = G.empty(n, 255), G.empty(n, 1024, dtype=__import__("torch").int32), G.empty(n, 1024), G.empty(n, 1280)
    blk.k_sample_pdf(G.dev(w), G.dev(z), None, n, g_cdf, g_ind, g_zs, g_zf)
    blk.synchronize()
    np.testing.assert_array_equal(g_cdf.cpu().numpy(), cdf)
    np.testing.assert_array_equal(g_ind.cpu().numpy(), ind)
    np.testing.assert_array_equal(g_zs.cpu().numpy(), zs)
    np.testing.assert_array_equal(g_zf.cpu().numpy(), zf)
    # end to end
    fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, debug=True)
    at_dev = o.forward(params, ro, rd, b, rd, rand={"z_fine_in": dbg["z_fine"]})
    np.testing.assert_allclose(fine, at_dev["rgb_fine"], rtol=0, atol=2e-4)
    ref = o.forward(params, ro, rd, b, rd)
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])
    assert (np.diff(dbg["z_fine"], axis=1) >= 0).all()
    np.testing.assert_allclose(fine, ref["rgb_fine"], rtol=0, atol=1e-3)
    lf, lc, dbg = blk.train_backward(ro, rd, b, rd, label, debug=True)
    r = o.train_grads(params, ro, rd, b, rd, label, rand={"z_fine_in": dbg["z_fine"]})
    assert lf == pytest.approx(r["loss_fine"], rel=1e-4) and lc == pytest.approx(r["loss_coarse"], rel=1e-4)
    H.assert_grads_close(c, blk.get_grads(), r["grads"], whole=1e-3)
    blk.close()


def test_empty_batch():
    c = H.CONFIGS["small_dir"]
    blk = G.make_block(c)
    blk.set_params(H.make_params(c, 1))
    e3, e2 = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32)
    assert blk.evaluate(e3, e3, e2, e3).shape == (0, 3)                       # renderToImage over no rays: nothing to do
    assert blk.render_u8(e3, e3, e2, e3).shape == (0, 3)
    with pytest.raises(NerfHipError):                                         # a training step needs rays: mean over 0 values
        blk.train_backward(e3, e3, e2, e3, e3)
    with pytest.raises(NerfHipError):
        blk.render_dev(0, 0, 0, 0, 0, 0)                                      # null device pointers
    blk.close()
