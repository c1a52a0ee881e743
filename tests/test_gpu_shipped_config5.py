"""Two run shapes that round 1 only executed inside bench.py, through the C ABI:

* the reference's SHIPPED configuration (runNerf.scala:24-56): SH colour head, no hierarchical pass, rawNoiseStd = 1,
  64 samples, 8x256, batchNum = 1024 -- render and one forward+loss+backward against the oracle, fp32 and bf16;
* BASELINE configs[4] TRAINING: forward-facing NDC rays (llff.scala:96-132; bounds (0, |d'|), getDataSet.scala:112),
  64 + 256 samples per ray, full 8x256 view-direction network -- gradients in fp32 (vs the oracle) and on the bf16-MFMA
  path (vs tests/bf16_ref.py, the bars of test_gradients_bf16)."""
import numpy as np
import pytest

import helpers as H
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available()
    import gpu_util
    return gpu_util


def _spans(c):
    off, spans = 0, []
    for _ in range(2 if c["useHierarchical"] else 1):
        for nm, r_, c_ in H.layout(c):
            spans.append((nm, off, r_ * c_)); off += r_ * c_
    return spans


def _check_bf16_grads(c, gr, rg):
    assert np.isfinite(gr).all()
    assert np.linalg.norm(gr - rg) / np.linalg.norm(rg) < 2e-2
    spans = _spans(c)
    big = max(np.linalg.norm(rg[o_:o_ + k]) for _, o_, k in spans)
    for nm, o_, k in spans:
        assert np.linalg.norm(gr[o_:o_ + k] - rg[o_:o_ + k]) <= 1e-1 * max(np.linalg.norm(rg[o_:o_ + k]), 0.05 * big), nm


@pytest.fixture(scope="module")
def shipped():
    c = H.CONFIGS["shipped"]
    assert c["useSH"] and not c["useHierarchical"] and c["rawNoiseStd"] == 1.0 and c["NSamples"] == 64 \
        and c["D"] == 8 and c["W"] == 256 and c["skips"] == [4] and c["posL"] == 10       # runNerf.scala:24-56
    n = 1024                                                                              # batchNum (:40)
    ro, rd, b, label = H.make_rays(n, 41)
    params = H.make_params(c, 9)
    noise = H.det_normal(n * 64, 77).reshape(n, 64)
    o = Oracle(c)
    ev = o.forward(params, ro, rd, b, rd, training=False)
    tr = o.train_grads(params, ro, rd, b, rd, label, rand={"noise_f": noise})
    return dict(c=c, n=n, ro=ro, rd=rd, b=b, label=label, params=params, noise=noise, ev=ev, tr=tr)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16"])
def test_shipped_config_render(G, shipped, dtype):
    """eval forward (no noise, nerfBlock.scala:21) of a 1024-ray batch: fp32 rgb within 2e-4 abs and depth/acc 1e-3 / 1e-5
    relative of the oracle, stratified depths bit-exact; bf16 within 1e-2 abs."""
    s = shipped
    blk = G.make_block(s["c"], max_rays=s["n"], dtype=dtype)
    blk.set_params(s["params"])
    fine, coarse, dbg = blk.forward(s["ro"], s["rd"], s["b"], s["rd"], training=False, debug=True)
    assert coarse is None
    np.testing.assert_array_equal(dbg["z_coarse"], s["ev"]["z_coarse"])
    if dtype != "bf16":   # fp32 and split-f16 kernels: one bar
        np.testing.assert_allclose(fine, s["ev"]["rgb_fine"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(dbg["depth"], s["ev"]["depth"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(dbg["acc"], s["ev"]["acc"], rtol=1e-5, atol=1e-6)
        # per-sample raw outputs of the SH head: identical inputs (bit-exact depths), so every sample is held to the
        # MLP-stage bar
        np.testing.assert_allclose(dbg["sigma_f"][:, :-1], s["ev"]["sigma_f"][:, :-1], rtol=2e-4, atol=2e-5 * np.abs(s["ev"]["sigma_f"]).max())
        np.testing.assert_allclose(dbg["craw_f"], s["ev"]["craw_f"], rtol=2e-4, atol=2e-5 * np.abs(s["ev"]["craw_f"]).max())
    else:
        np.testing.assert_allclose(fine, s["ev"]["rgb_fine"], rtol=0, atol=1e-2)
    u8 = blk.render_u8(s["ro"], s["rd"], s["b"], s["rd"])
    np.testing.assert_array_equal(u8, Oracle(s["c"]).to_u8(fine))          # .mul(255).toType(UINT8) of ITS OWN floats
    blk.close()


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_shipped_config_train_backward_f32(G, shipped, dtype):
    """(fp32 bars; dtype f16x3 = the split-f16 training kernels, held to the same ones)  forward(training, noise on the raw density) + L2 + backward with the noise tensor passed in: loss 1e-4 relative,
    whole gradient < 1e-3 relative L2 of the oracle's, per tensor norm 1e-2; and the library-drawn noise path runs the
    same batch to a finite loss of the same size."""
    s = shipped
    c = s["c"]
    blk = G.make_block(c, max_rays=s["n"], dtype=dtype)
    blk.set_params(s["params"])
    lf, lc = blk.train_backward(s["ro"], s["rd"], s["b"], s["rd"], s["label"], rand={"noise_f": s["noise"]})
    gr, rg = blk.get_grads(), s["tr"]["grads"]
    assert lf == pytest.approx(s["tr"]["loss_fine"], rel=1e-4) and lc == 0.0
    assert np.linalg.norm(gr - rg) / np.linalg.norm(rg) < 1e-3
    for nm, o_, k in _spans(c):
        assert np.linalg.norm(gr[o_:o_ + k]) == pytest.approx(np.linalg.norm(rg[o_:o_ + k]), rel=1e-2), nm
    lp, _ = blk.train_backward(s["ro"], s["rd"], s["b"], s["rd"], s["label"])      # Philox noise (nerfhip_aux NULL)
    assert np.isfinite(lp) and lp == pytest.approx(lf, rel=0.2) and lp != lf
    assert np.isfinite(blk.get_grads()).all()
    blk.optimizer_step()
    assert np.isfinite(blk.get_params()).all()
    blk.close()


def test_shipped_config_train_backward_bf16(G, shipped):
    import bf16_ref
    s = shipped
    c = s["c"]
    blk = G.make_block(c, max_rays=s["n"], dtype="bf16")
    blk.set_params(s["params"])
    lf, _ = blk.train_backward(s["ro"], s["rd"], s["b"], s["rd"], s["label"], rand={"noise_f": s["noise"]})
    rlf, _, rg = bf16_ref.train_grads(c, s["params"], s["ro"], s["rd"], s["b"], s["label"], {"noise_f": s["noise"]})
    assert lf == pytest.approx(rlf, rel=1e-2)
    _check_bf16_grads(c, blk.get_grads(), rg)
    assert np.linalg.norm(blk.get_grads() - s["tr"]["grads"]) / np.linalg.norm(s["tr"]["grads"]) < 0.3   # sanity vs fp32
    blk.close()


@pytest.fixture(scope="module")
def config5():
    c = dict(H.CONFIGS["full"], NImportance=256)
    Hh, Ww, f = 8, 12, 10.0
    c2w = np.array([[1, 0, 0, 0.1], [0, 1, 0, -0.05], [0, 0, 1, 0.2]], np.float32)
    o = Oracle(c)
    ro, rd, b = o.gen_rays(Hh, Ww, f, c2w, 0.0, 1.0, True)          # NDC rays, bounds (0, |d'|)
    n = Hh * Ww
    assert (b[:, 0] == 0).all()
    label = (0.5 + 0.5 * np.sin(3 * rd + ro)).astype(np.float32)
    params = H.make_params(c, 5)
    return dict(c=c, n=n, ro=ro, rd=rd, b=b, label=label, params=params)


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_config5_training_f32(G, config5, dtype):
    """configs[4] shapes in fp32 (and on the fp32-grade split-f16 kernels) against the oracle: the bars of
    test_gradients_full -- at the device's fine depths losses 1e-5, whole gradient < 1e-4, every tensor within 5e-3 of its
    norm; at the oracle's own fine depths losses 1e-3, whole gradient < 5e-3, no fine depth off by more than a coarse bin."""
    s = config5
    c = s["c"]
    blk = G.make_block(c, max_rays=s["n"], dtype=dtype)
    blk.set_params(s["params"])
    lf, lc, dbg = blk.train_backward(s["ro"], s["rd"], s["b"], s["rd"], s["label"], debug=True)
    assert dbg["z_fine"].shape == (s["n"], 320)
    o = Oracle(c)
    gr = blk.get_grads()
    at_dev = o.train_grads(s["params"], s["ro"], s["rd"], s["b"], s["rd"], s["label"], rand={"z_fine_in": dbg["z_fine"]})
    assert lf == pytest.approx(at_dev["loss_fine"], rel=1e-5) and lc == pytest.approx(at_dev["loss_coarse"], rel=1e-5)
    H.assert_grads_close(c, gr, at_dev["grads"], whole=1e-4)
    ref = o.train_grads(s["params"], s["ro"], s["rd"], s["b"], s["rd"], s["label"])
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])
    assert lf == pytest.approx(ref["loss_fine"], rel=1e-3) and lc == pytest.approx(ref["loss_coarse"], rel=1e-3)
    H.assert_grads_close(c, gr, ref["grads"], whole=5e-3, per_tensor=2e-2)   # one moved fine sample (37 rays: 1.1e-3 measured)
    bin_w = np.diff(ref["z_coarse"], axis=1).max(axis=1, keepdims=True)
    assert (np.abs(dbg["z_fine"] - ref["z_fine"]) <= bin_w).all()
    blk.close()


def test_config5_training_bf16(G, config5):
    """configs[4] as BASELINE names it: bf16-MFMA training, NDC rays, 256 fine samples -- against the bf16 emulation."""
    import bf16_ref
    s = config5
    c = s["c"]
    blk = G.make_block(c, max_rays=s["n"], dtype="bf16")
    blk.set_params(s["params"])
    lf, lc = blk.train_backward(s["ro"], s["rd"], s["b"], s["rd"], s["label"])
    rlf, rlc, rg = bf16_ref.train_grads(c, s["params"], s["ro"], s["rd"], s["b"], s["label"], None)
    assert lf == pytest.approx(rlf, rel=1e-2) and lc == pytest.approx(rlc, rel=1e-2)
    _check_bf16_grads(c, blk.get_grads(), rg)
    # and a few optimizer steps reduce the loss
    first = lf
    for _ in range(5):
        lf, _ = blk.train_step(s["ro"], s["rd"], s["b"], s["rd"], s["label"])
    assert np.isfinite(lf) and lf < first
    blk.close()
