"""GPU parity of the render path: every stage of libnerfhip against the CPU
oracle on identical inputs, through the C ABI.  Bit-exact where the arithmetic is elementwise fp32 or
integer (sample depths, CDF, bin indices, merged fine depths, ray generation, uint8 rows); tolerances
stated at each assert otherwise."""
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


@pytest.mark.parametrize("name", ["full", "small_opts", "small_flat"])
def test_stratified_bit_exact(G, name):
    import torch
    c = H.CONFIGS[name]
    n = 1000
    _, _, bounds, _ = H.make_rays(n, 3)
    t_rand = H.det_uniform(n * c["NSamples"], 11).reshape(n, -1) if c["perturb"] else None
    blk = G.make_block(c)
    z = G.empty(n, c["NSamples"])
    blk.k_stratified(G.dev(bounds), G.dev(t_rand) if t_rand is not None else None, n, z)
    blk.synchronize()
    ref = Oracle(c).stratified(bounds, t_rand)
    np.testing.assert_array_equal(z.cpu().numpy(), ref)
    blk.close()


@pytest.mark.parametrize("name,N,NI", [("full", 64, 128), ("small_dir", 16, 24), ("small_opts", 16, 24),
                                       ("full", 64, 256), ("full", 100, 61)])
def test_sample_pdf_bit_exact(G, name, N, NI):
    """CDF, bin indices, samples and merged/sorted fine depths are bit-identical to the oracle on the same
    (w, z), including peaked, flat and zero weights and u ties."""
    c = H.cfg(**{**H.CONFIGS[name], "NSamples": N, "NImportance": NI})
    n = 777
    o = Oracle(c)
    _, _, bounds, _ = H.make_rays(n, 5)
    z = o.stratified(bounds) if not c["perturb"] else o.stratified(bounds, H.det_uniform(n * N, 2).reshape(n, N))
    w = H.det_uniform(n * N, 7).reshape(n, N) ** 4
    w[: n // 4] *= 1e-4                       # nearly empty rays -> den < 1e-5 branch
    w[n // 4: n // 2, N // 3] = 0.9           # one dominant bin
    w[-5:] = 0.0                              # all-zero weights: uniform cdf from the +1e-5 term
    w = (w / np.maximum(w.sum(1, keepdims=True), 1.0)).astype(np.float32)
    u_rand = H.det_uniform(n * NI, 9).reshape(n, NI) if c["perturb"] else None
    blk = G.make_block(c)
    cdf, ind, zs, zf = G.empty(n, N - 1), G.empty(n, NI, dtype=__import__("torch").int32), G.empty(n, NI), G.empty(n, N + NI)
    blk.k_sample_pdf(G.dev(w), G.dev(z), G.dev(u_rand) if u_rand is not None else None, n, cdf, ind, zs, zf)
    blk.synchronize()
    rc, ri, rs, rf = o.sample_pdf(w, z, u_rand)
    np.testing.assert_array_equal(cdf.cpu().numpy(), rc)
    np.testing.assert_array_equal(ind.cpu().numpy(), ri)
    np.testing.assert_array_equal(zs.cpu().numpy(), rs)
    np.testing.assert_array_equal(zf.cpu().numpy(), rf)
    blk.close()


def test_sample_pdf_unsorted_depths(G):
    """The reference sorts concat(samples, z) whatever z is (nerfBlock.scala:101).  The kernel merges when z ascends
    and falls back to a full sort per ray otherwise: rays with shuffled / descending / constant z, mixed with
    ordinary ones in the same workgroup, still match the oracle bit for bit."""
    import torch
    c = H.cfg(**{**H.CONFIGS["full"], "NSamples": 64, "NImportance": 128})
    n, N, NI = 203, 64, 128
    o = Oracle(c)
    _, _, bounds, _ = H.make_rays(n, 6)
    z = o.stratified(bounds).copy()
    rng = np.random.default_rng(3)
    for r in range(0, n, 3):
        z[r] = z[r][rng.permutation(N)]          # shuffled
    z[1] = z[1][::-1]                             # descending
    z[4] = z[4][0]                                # constant
    w = (H.det_uniform(n * N, 11).reshape(n, N) ** 2).astype(np.float32)
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    blk = G.make_block(c)
    ind, zf = G.empty(n, NI, dtype=torch.int32), G.empty(n, N + NI)
    blk.k_sample_pdf(G.dev(w), G.dev(z), None, n, None, ind, None, zf)
    blk.synchronize()
    _, ri, _, rf = o.sample_pdf(w, z, None)
    np.testing.assert_array_equal(ind.cpu().numpy(), ri)
    np.testing.assert_array_equal(zf.cpu().numpy(), rf)
    blk.close()


def test_sample_pdf_golden(G):
    import torch
    for case in ("full_eval", "small_opts_eval"):
        name = case.rsplit("_", 1)[0]
        c = H.CONFIGS[name]
        g = load(case)
        n, N, NI = g["w_coarse"].shape[0], c["NSamples"], c["NImportance"]
        blk = G.make_block(c)
        ind, zf = G.empty(n, NI, dtype=torch.int32), G.empty(n, N + NI)
        u = g.get("rand_u_rand")
        blk.k_sample_pdf(G.dev(g["w_coarse"]), G.dev(g["z_coarse"]), G.dev(u) if u is not None else None, n, None, ind,
                         None, zf)
        blk.synchronize()
        np.testing.assert_array_equal(ind.cpu().numpy(), g["ind"])
        np.testing.assert_array_equal(zf.cpu().numpy(), g["z_fine"])
        blk.close()


@pytest.mark.parametrize("S,white", [(64, True), (192, True), (320, False), (17, True), (128, False), (250, True), (400, True), (2, True)])
def test_composite_fwd(G, S, white):
    """fp32 compositing: weights / rgb / depth / acc within 1e-5 relative of the oracle (north_star)."""
    c = H.cfg(whiteBkgd=white)
    n = 1500
    o = Oracle(c)
    sigma = (H.det_normal(n * S, 3).reshape(n, S) * 3).astype(np.float32)
    sigma[:50] = -1.0                                # empty rays
    sigma[50:100] = 50.0                             # opaque at the first sample
    craw = (H.det_normal(n * S * 3, 4).reshape(n, S, 3) * 2).astype(np.float32)
    z = np.sort(H.det_uniform(n * S, 5, 2.0, 6.0).reshape(n, S), axis=1)
    blk = G.make_block(c)
    w, rgb, depth, acc = G.empty(n, S), G.empty(n, 3), G.empty(n), G.empty(n)
    blk.k_composite_fwd(G.dev(sigma), None, G.dev(craw), G.dev(z), n, S, w, rgb, depth, acc)
    blk.synchronize()
    rw, rrgb, rdepth, racc = o.composite(sigma, craw, z)
    np.testing.assert_allclose(w.cpu().numpy(), rw, rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), rrgb, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(depth.cpu().numpy(), rdepth, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(acc.cpu().numpy(), racc, rtol=1e-5, atol=1e-6)
    blk.close()


MLP_CASES = ["small_dir_eval", "small_sh_eval", "small_nodir_train", "small_flat_eval", "full_eval"]


# every fp32 render test also runs on the split-f16 kernels (dtype "f16x3": three f16 MFMAs per product, fp32-grade),
# at the SAME tolerances; W = 32 networks exist only in fp32
DTYPES = ["f32", "f16x3"]


# (case, dtype) pairs of the MLP stage test: the split-f16 kernels need W = 64, 128 or 256, so the W = 32 golden is fp32 only
MLP_STAGE = [(cs, dt) for cs in MLP_CASES for dt in DTYPES
             if not (dt == "f16x3" and H.CONFIGS[cs.rsplit("_", 1)[0]]["W"] == 32)]


@pytest.mark.parametrize("case,dtype", MLP_STAGE)
def test_mlp_forward_stage(G, case, dtype):
    """Fused posenc+MLP kernel on the golden's own fine depths vs the oracle and the golden (fp32 MFMA:
    only the GEMM summation order and 1-2 ulp sin/cos differ; split-f16: the same bar)."""
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    o = Oracle(c)
    params = H.make_params(c, int(g["seed"]))
    blk = G.make_block(c, dtype=dtype)
    blk.set_params(params)
    n, S = g["z_fine"].shape
    net = 1 if c["useHierarchical"] else 0
    sig, craw = G.empty(n, S), G.empty(n, S, 3)
    blk.k_mlp_forward(net, G.dev(g["rays_o"]), G.dev(g["rays_d"]), G.dev(g["rays_d"]), G.dev(g["z_fine"]), n, S, sig, craw)
    blk.synchronize()
    npc = o.net_param_count()
    rs, rc = o.mlp_forward(params[npc * net:npc * (net + 1)], False, True, g["rays_o"], g["rays_d"], g["rays_d"], g["z_fine"])
    sig, craw = sig.cpu().numpy(), craw.cpu().numpy()
    np.testing.assert_allclose(sig[:, :S - 1], rs[:, :S - 1], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(craw, rc, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(sig[:, :S - 1], g["sigma_f"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(craw, g["craw_f"], rtol=2e-4, atol=2e-5)
    blk.close()


KORDER = {
    "w256_dir": H.cfg(D=4, W=256, skips=[1], posL=0, dirL=0, NSamples=16, NImportance=0, useHierarchical=False),
    "w128_sh": H.cfg(D=3, W=128, skips=[0], posL=0, dirL=0, NSamples=16, NImportance=0, useHierarchical=False, useSH=True),
    "w64_nodir": H.cfg(D=5, W=64, skips=[2], posL=0, dirL=0, NSamples=16, NImportance=0, useHierarchical=False, useDir=False),
    "w32_dir": H.cfg(D=3, W=32, skips=[], posL=0, dirL=0, NSamples=16, NImportance=0, useHierarchical=False),
    "full_pos0": H.cfg(posL=0, dirL=0, NImportance=0, useHierarchical=False),
}


@pytest.mark.parametrize("name", sorted(KORDER))
def test_mlp_stage_bit_exact_in_kernel_order(G, name):
    """fp32 MFMA == an fmaf chain, bitwise (MI355X_MICROARCH.md).  With posL = dirL = 0 the network has no sin/cos, so
    the fused kernel must reproduce, BIT FOR BIT, the oracle evaluated in the kernel's own summation order
    (orc_mlp_forward_korder: k = 32kt + 8q + 4h + j, heads as two half-chains): raw density and raw colour of every
    sample.  Any mis-indexed weight, dropped k or wrong bias shows up here as a non-zero difference."""
    c = KORDER[name]
    o = Oracle(c)
    n, S = 70, 13
    ro, rd, b, _ = H.make_rays(n, 9)
    z = np.sort(H.det_uniform(n * S, 5, 2.0, 6.0).reshape(n, S), axis=1)
    params = H.make_params(c, 8)
    blk = G.make_block(c)
    blk.set_params(params)
    sig, craw = G.empty(n, S), G.empty(n, S, 3)
    blk.k_mlp_forward(0, G.dev(ro), G.dev(rd), G.dev(rd), G.dev(z), n, S, sig, craw)
    blk.synchronize()
    rs, rc = o.mlp_forward_korder(params, True, ro, rd, rd, z)
    np.testing.assert_array_equal(sig.cpu().numpy()[:, :S - 1], rs[:, :S - 1])
    np.testing.assert_array_equal(craw.cpu().numpy(), rc)
    blk.close()


@pytest.mark.parametrize("case", MLP_CASES)
def test_mlp_forward_stage_tight(G, case):
    """The golden networks (posL > 0) against the kernel-order oracle: what is left is the 1-2 ulp between the device's
    sincosf and libm's sinf/cosf, carried through the layers -- every sample within 2e-6 of the output scale
    (100x tighter than the bar a different GEMM order forces in test_mlp_forward_stage)."""
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    o = Oracle(c)
    params = H.make_params(c, int(g["seed"]))
    blk = G.make_block(c)
    blk.set_params(params)
    n, S = g["z_fine"].shape
    net = 1 if c["useHierarchical"] else 0
    sig, craw = G.empty(n, S), G.empty(n, S, 3)
    blk.k_mlp_forward(net, G.dev(g["rays_o"]), G.dev(g["rays_d"]), G.dev(g["rays_d"]), G.dev(g["z_fine"]), n, S, sig, craw)
    blk.synchronize()
    npc = o.net_param_count()
    rs, rc = o.mlp_forward_korder(params[npc * net:npc * (net + 1)], True, g["rays_o"], g["rays_d"], g["rays_d"], g["z_fine"])
    sig, craw = sig.cpu().numpy(), craw.cpu().numpy()
    assert np.abs(sig[:, :S - 1] - rs[:, :S - 1]).max() <= 2e-6 * np.abs(rs).max()
    assert np.abs(craw - rc).max() <= 2e-6 * np.abs(rc).max()
    blk.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_mlp_forward_ragged_tile(G, dtype):
    """n*S not a multiple of the 128-sample workgroup tile; coarse network without colour head."""
    c = H.CONFIGS["small_dir"]
    o = Oracle(c)
    n, S = 37, 19
    ro, rd, b, _ = H.make_rays(n, 8)
    z = np.sort(H.det_uniform(n * S, 5, 2.0, 6.0).reshape(n, S), axis=1)
    params = H.make_params(c, 3)
    blk = G.make_block(c, dtype=dtype)
    blk.set_params(params)
    sig = G.empty(n, S)
    blk.k_mlp_forward(0, G.dev(ro), G.dev(rd), G.dev(rd), G.dev(z), n, S, sig, None)
    blk.synchronize()
    rs, _ = o.mlp_forward(params[:o.net_param_count()], True, False, ro, rd, rd, z)
    np.testing.assert_allclose(sig.cpu().numpy()[:, :S - 1], rs[:, :S - 1], rtol=2e-4, atol=2e-5)
    blk.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_mlp_forward_far_positions(G, dtype):
    """Encoding arguments beyond the short sin/cos routine's range (|p| * 2^(posL-1) > 8192, mlp_common.hip.h: kTrigMax):
    waves whose samples are all far, and waves that mix far and near samples, take libm's functions; same bar."""
    c = H.CONFIGS["small_dir"]
    o = Oracle(c)
    n, S = 96, 16
    ro, rd, b, _ = H.make_rays(n, 11)
    ro = ro.copy()
    ro[:40] *= 2000.0          # |p| ~ 8000: arguments up to 6.4e4
    ro[40:64:2] *= 2000.0      # every other ray: far and near samples inside one wave
    z = np.sort(H.det_uniform(n * S, 6, 2.0, 6.0).reshape(n, S), axis=1)
    params = H.make_params(c, 4)
    blk = G.make_block(c, dtype=dtype)
    blk.set_params(params)
    sig, craw = G.empty(n, S), G.empty(n, S, 3)
    blk.k_mlp_forward(1, G.dev(ro), G.dev(rd), G.dev(rd), G.dev(z), n, S, sig, craw)
    blk.synchronize()
    npc = o.net_param_count()
    rs, rc = o.mlp_forward(params[npc:2 * npc], False, True, ro, rd, rd, z)
    # raw positions of 8000 enter layer 0 as they are: outputs scale with them, so the bar is relative to the output scale
    np.testing.assert_allclose(sig.cpu().numpy()[:, :S - 1], rs[:, :S - 1], rtol=2e-4, atol=2e-5 * max(1.0, np.abs(rs).max()))
    np.testing.assert_allclose(craw.cpu().numpy(), rc, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(rc).max()))
    blk.close()


E2E = ["small_dir_eval", "small_sh_eval", "small_flat_eval", "small_opts_eval", "full_eval"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", E2E)
def test_render_end_to_end(G, case, dtype):
    """nerfhip_render (host pointers) vs oracle and golden.  Sample depths bit-exact; per-sample tensors as a
    fraction within tolerance (a 1-ulp depth difference is amplified by the 2^(posL-1) encoding, see
    test_oracle_golden.py); composited rgb within 2e-4 abs, depth 2e-4 rel."""
    name = case.rsplit("_", 1)[0]
    c = H.CONFIGS[name]
    g = load(case)
    o = Oracle(c)
    params = H.make_params(c, int(g["seed"]))
    rand = {k[5:]: v for k, v in g.items() if k.startswith("rand_")}
    blk = G.make_block(c, dtype=dtype)
    blk.set_params(params)
    fine, coarse, dbg = blk.forward(g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], training=False, rand=rand, debug=True)
    ref = o.forward(params, g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], training=False, rand=rand)
    assert coarse is None
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])
    S = dbg["z_fine"].shape[1]
    if c["useHierarchical"]:
        N = c["NSamples"]
        np.testing.assert_allclose(dbg["sigma_c"][:, :N - 1], ref["sigma_c"][:, :N - 1], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(dbg["w_coarse"], ref["w_coarse"], rtol=2e-4, atol=2e-6)
        # The fine depths are a continuous function of the coarse weights (which differ from the oracle's in the last
        # bits: GEMM order) except at the `den < 1e-5 -> 1` branch of samplePdf, where a last-ulp weight difference
        # moves an isolated sample: >= 99 % within 1e-4 relative; every row ascends and keeps the coarse end points.
        assert G.close_frac(dbg["z_fine"], ref["z_fine"], 1e-4, 1e-5) > 0.99
        # ... and the moved ones are bounded: by rank, a moved sample lands inside a neighbouring bin, so no merged depth is
        # further from the oracle's than one coarse bin of its ray
        bin_w = np.diff(ref["z_coarse"], axis=1).max(axis=1, keepdims=True)
        assert (np.abs(dbg["z_fine"] - ref["z_fine"]) <= bin_w).all()
        assert (np.diff(dbg["z_fine"], axis=1) >= 0).all()
        assert (dbg["z_fine"][:, 0] == ref["z_coarse"][:, 0]).all() and (dbg["z_fine"][:, -1] == ref["z_coarse"][:, -1]).all()
    # PER SAMPLE, no exemptions: the fine network's raw density and colour of EVERY sample against the oracle
    # evaluated (in the kernel's summation order) at the depth the library actually used -- a last-bit depth
    # difference is amplified by the 2^(posL-1) encoding, so comparing at the oracle's own depths would test the
    # sampler, not the network.  2e-6 of the output scale (sin/cos ulps), as test_mlp_forward_stage_tight.
    npc = o.net_param_count()
    net = 1 if c["useHierarchical"] else 0
    if dtype == "f32":
        ks, kc = o.mlp_forward_korder(params[npc * net:npc * (net + 1)], True, g["rays_o"], g["rays_d"], g["rays_d"], dbg["z_fine"])
        assert np.abs(dbg["sigma_f"][:, :S - 1] - ks[:, :S - 1]).max() <= 2e-6 * np.abs(ks).max()
        assert np.abs(dbg["craw_f"] - kc).max() <= 2e-6 * np.abs(kc).max()
    else:   # split-f16 sums in its own order: every sample at the MLP-stage bar of the fp32 kernels
        ks, kc = o.mlp_forward(params[npc * net:npc * (net + 1)], False, True, g["rays_o"], g["rays_d"], g["rays_d"], dbg["z_fine"])
        np.testing.assert_allclose(dbg["sigma_f"][:, :S - 1], ks[:, :S - 1], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(dbg["craw_f"], kc, rtol=2e-4, atol=2e-5)
    for r in (ref, g):
        np.testing.assert_allclose(fine, r["rgb_fine"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(dbg["depth"], r["depth"], rtol=1e-3)  # end to end: an isolated moved sample
        np.testing.assert_allclose(dbg["acc"], r["acc"], rtol=1e-5, atol=1e-6)
    blk.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_render_chunking_and_u8(G, dtype):
    """n > max_rays is processed in slices; uint8 rows are *255, clamp, truncate (runNerf.scala:194)."""
    c = H.CONFIGS["small_sh"]
    o = Oracle(c)
    n = 300
    ro, rd, b, _ = H.make_rays(n, 12)
    params = H.make_params(c, 4)
    blk = G.make_block(c, max_rays=128, dtype=dtype)
    blk.set_params(params)
    fine = blk.evaluate(ro, rd, b, rd)
    ref = o.forward(params, ro, rd, b, rd)["rgb_fine"]
    np.testing.assert_allclose(fine, ref, rtol=0, atol=2e-4)
    u8 = blk.render_u8(ro, rd, b, rd)
    np.testing.assert_array_equal(u8, o.to_u8(fine))
    img = blk.render_to_image(ro[:288].reshape(2, 12, 12, 3), rd[:288].reshape(2, 12, 12, 3), b[:288].reshape(2, 12, 12, 2))
    np.testing.assert_array_equal(img.reshape(-1, 3), u8[:288])
    blk.close()


@pytest.mark.parametrize("ndc", [False, True])
def test_gen_rays_bit_exact(G, ndc):
    """Ray -> pixel mapping and ray arithmetic are bit-exact vs the oracle (getDataSet.scala:12-21,105-112)."""
    c = H.CONFIGS["small_dir"]
    o = Oracle(c)
    Hh, Ww, f = 37, 53, 61.5
    th = 0.3
    c2w = np.array([[np.cos(th), 0, np.sin(th), 0.4], [0, 1, 0, -0.2], [-np.sin(th), 0, np.cos(th), 3.5]], np.float32)
    blk = G.make_block(c)
    ro, rd, b = G.empty(Hh * Ww, 3), G.empty(Hh * Ww, 3), G.empty(Hh * Ww, 2)
    blk.k_gen_rays(Hh, Ww, f, c2w, 2.0, 6.0, ndc, ro, rd, b)
    blk.synchronize()
    eo, ed, eb = o.gen_rays(Hh, Ww, f, c2w, 2.0, 6.0, ndc)
    np.testing.assert_array_equal(ro.cpu().numpy(), eo)
    np.testing.assert_array_equal(rd.cpu().numpy(), ed)
    np.testing.assert_array_equal(b.cpu().numpy(), eb)
    blk.close()


def test_errors_are_loud(G):
    from nerf_amd import NerfHipError
    with pytest.raises(NerfHipError):
        G.make_block(H.cfg(W=100))
    blk = G.make_block(H.CONFIGS["small_dir"], max_rays=16)
    bad = np.zeros(3, np.float32)
    rc = blk._lib.nerfhip_set_params(blk._h, bad.ctypes.data, 3)   # straight through the C ABI
    assert rc == -1 and b"wrong size" in blk._lib.nerfhip_last_error(blk._h)
    rc = blk._lib.nerfhip_render_dev(blk._h, None, None, None, None, 4, None, None)
    assert rc == -1
    blk.close()


@pytest.mark.parametrize("ndc", [False, True])
def test_render_pose_matches_ray_level_render(G, ndc):
    """nerfhip_render_pose_u8 (rays made on the device from the pose, rendered in max_rays blocks) == render_u8 on the
    rays of k_gen_rays, byte for byte; pixel (j, i) at [j, i]; and within one grey level of the oracle's image."""
    c = H.CONFIGS["small_dir"]
    o = Oracle(c)
    Hh, Ww, f = 23, 31, 40.0
    th = 0.4
    c2w = np.array([[np.cos(th), 0, np.sin(th), 0.3], [0, 1, 0, -0.1], [-np.sin(th), 0, np.cos(th), 3.8]], np.float32)
    blk = G.make_block(c, max_rays=200)          # 713 rays -> 4 blocks, the last one short
    blk.set_params(H.make_params(c, 4))
    near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
    img = blk.render_pose(Hh, Ww, f, c2w, near, far, ndc)
    assert img.shape == (Hh, Ww, 3) and img.dtype == np.uint8
    eo, ed, eb = o.gen_rays(Hh, Ww, f, c2w, near, far, ndc)
    np.testing.assert_array_equal(img.reshape(-1, 3), blk.render_u8(eo, ed, eb, ed))
    ref = o.to_u8(o.forward(H.make_params(c, 4), eo, ed, eb, ed)["rgb_fine"]).reshape(Hh, Ww, 3)
    assert np.abs(img.astype(int) - ref.astype(int)).max() <= 1
    blk.close()
