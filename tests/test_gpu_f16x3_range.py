"""dtype f16x3 range guard (include/nerfhip.h): the split-f16 operands are f16 pairs of 64*w and of the activations, so a
weight with |w| >= 1023.5 and an activation beyond 65504 have no representation.  Both are REJECTED -- NERFHIP_E_INVALID
with the offending tensor named, or a device flag that the next synchronising call reports -- instead of inf pixels; the
same weights on dtype f32 are fine.  No counterpart in the reference (its arithmetic is fp32 throughout)."""
import numpy as np
import pytest

import gpu_util as G
import helpers as H
from nerf_amd import NerfHipError

pytestmark = pytest.mark.gpu


def _spans(c):
    return {(net, nm): (off, k) for net, nm, off, k in H.param_spans(c)}


def test_set_params_rejects_weights_beyond_the_split_f16_range():
    c = H.CONFIGS["small_dir"]
    ro, rd, b, _ = H.make_rays(32, 3)
    good = H.make_params(c, 1)
    bad = good.copy()
    off, k = _spans(c)[(1, "w2")]
    bad[off + 5] = 2000.0
    blk = G.make_block(c, dtype="f16x3")
    with pytest.raises(NerfHipError) as e:
        blk.set_params(bad)
    msg = str(e.value)
    assert "network 1" in msg and "trunk layer 2 weight" in msg and str(off + 5) in msg, msg
    # the handle stays usable: good weights are accepted and render finite pixels
    blk.set_params(good)
    assert np.isfinite(blk.evaluate(ro, rd, b, rd)).all()
    # NaN weights are caught by the same test
    bad = good.copy()
    off, k = _spans(c)[(0, "wf")]
    bad[off] = np.nan
    with pytest.raises(NerfHipError) as e:
        blk.set_params(bad)
    assert "network 0" in str(e.value) and "colour head first matrix" in str(e.value)
    blk.close()
    # dtype f32 takes the same weights (its operands are fp32)
    f = G.make_block(c, dtype="f32")
    bad = good.copy(); bad[_spans(c)[(1, "w2")][0] + 5] = 2000.0
    f.set_params(bad)
    assert np.isfinite(f.evaluate(ro, rd, b, rd)).all()
    f.close()


def test_activation_overflow_is_reported_not_rendered():
    """weights inside the range whose ACTIVATIONS leave it: every trunk layer amplifies by ~20, so by layer 3 the
    pre-activations pass 65504 and the f16 high halves become inf"""
    c = H.CONFIGS["small_dir"]
    ro, rd, b, label = H.make_rays(32, 3)
    p = H.make_params(c, 1)
    for (net, nm), (off, k) in _spans(c).items():
        if nm in ("w0", "w1", "w2", "w3"):
            p[off:off + k] *= 60.0
    assert np.abs(p).max() < 1000.0
    blk = G.make_block(c, dtype="f16x3")
    blk.set_params(p)                      # the weights themselves are representable
    with pytest.raises(NerfHipError) as e:
        blk.evaluate(ro, rd, b, rd)
    assert "activation" in str(e.value) and "65504" in str(e.value)
    # asynchronous entry point: the flag is raised on the device and reported by the next synchronising call
    out = G.empty(32, 3)
    blk.render_dev(G.dev(ro), G.dev(rd), G.dev(b), G.dev(rd), 32, out)
    with pytest.raises(NerfHipError):
        blk.synchronize()
    blk.synchronize()                      # reported once, then cleared
    with pytest.raises(NerfHipError):      # the training forward is the same kernel
        blk.train_backward(ro, rd, b, rd, label)
    blk.set_params(H.make_params(c, 1))
    assert np.isfinite(blk.evaluate(ro, rd, b, rd)).all()
    blk.close()
    f = G.make_block(c, dtype="f32")       # fp32 has the range for it
    f.set_params(p)
    assert np.isfinite(f.evaluate(ro, rd, b, rd)).all()
    f.close()


def test_optimizer_step_leaving_the_range_is_reported():
    """a step that carries a weight out of the range: optimizer_step is asynchronous, the repack behind it raises the
    flag and the next synchronising call names the tensor"""
    c = H.CONFIGS["small_dir"]
    ro, rd, b, label = H.make_rays(32, 5)
    p = H.make_params(c, 2)
    off, k = _spans(c)[(1, "w1")]
    p[off + 7] = 1023.4                    # just inside
    blk = G.make_block(c, dtype="f16x3", lrate=50.0)
    blk.set_params(p)
    with pytest.raises(NerfHipError) as e:
        for _ in range(4):
            blk.train_backward(ro, rd, b, rd, label)
            blk.optimizer_step()
            blk.synchronize()
    assert "weight" in str(e.value) or "activation" in str(e.value)
    blk.close()
