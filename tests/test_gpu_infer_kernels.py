"""Since round 3 the bf16 and split-f16 paths run TWO forward kernels: inference on the 16x16x32 MFMA shape
(csrc/mlp_fwd_bf16_t16.hip, csrc/mlp_fwd_f16x3_t16.hip, own weight-fragment order) and the training forward on the 32x32x16
shape (csrc/mlp_fwd_bf16.hip, csrc/mlp_fwd_f16x3.hip).  They must be the same function of the same parameters: the training
forward's raw density / colour of every fine sample (debug outputs of nerfhip_train_backward, noise off) against the inference
kernel evaluated at the SAME depths (nerfhip_k_mlp_forward) -- no sampler in between, so no `den < 1e-5` discontinuity.
split-f16: both are fp32-grade, they differ by the order in which an MFMA adds its products (measured <= 1.3e-6 of the output
scale; bar 1e-5).  bf16: two roundings of the same sums can fall on different bf16 values and move an isolated sample by a few
1e-3 of the scale (measured: max 5.3e-3, mean 1.2e-5; bar: 2e-2 of the scale for every sample, 1e-4 on average) -- the bf16
bars elsewhere (1e-2 abs on rgb) are against the fp32 oracle.  Also: the parameters the inference streams are packed from follow an optimizer step (the split-f16 streams
are packed on demand).
The same comparison over the 1 987 random configurations with W >= 64 of tests/test_gpu_fuzz.py's generator, seeds 0-2499
(scripts/fuzz_infer_kernels.py, profiles/r3_fuzz_infer_kernels.txt): none over these bars -- split-f16 worst 4.4e-6, bf16 worst
1.2e-2 with mean 6.4e-5."""
import numpy as np
import pytest

import gpu_util as G
import helpers as H

pytestmark = pytest.mark.gpu

SHAPES = {
    "full": H.CONFIGS["full"],
    "shipped_sh": H.CONFIGS["shipped"],
    "w64_two_skips": H.cfg(D=6, W=64, skips=[1, 3], posL=4, dirL=1, NSamples=16, NImportance=16, useSH=True),
    "w128_nodir": H.cfg(D=3, W=128, skips=[1], posL=5, dirL=2, NSamples=17, NImportance=9, useDir=False, whiteBkgd=False),
}


def _stage(blk, c, ro, rd, b, label):
    """training step's forward (32x32x16 kernel: nerfhip_train_backward; nerfhip_forward_train runs the inference kernel)
    -> (sigma, craw, z) of the fine pass; inference kernel at the same z"""
    hier = c["useHierarchical"]
    _, _, dbg = blk.train_backward(ro, rd, b, rd, label, debug=True)
    z = dbg["z_fine"] if hier else dbg["z_coarse"]
    sig_t = dbg["sigma_f"]
    craw_t = dbg["craw_f"]
    n, S = z.shape
    sig = G.empty(n, S); craw = G.empty(n, S, 3)
    blk.k_mlp_forward(1 if hier else 0, G.dev(ro), G.dev(rd), G.dev(rd), G.dev(z), n, S, sig, craw)
    blk.synchronize()
    return sig_t, craw_t, sig.cpu().numpy(), craw.cpu().numpy()


@pytest.mark.parametrize("dtype", ["f16x3", "bf16"])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_inference_kernel_equals_training_forward(shape, dtype):
    c = dict(SHAPES[shape], rawNoiseStd=0.0, perturb=False)
    n = 70   # not a multiple of any tile
    ro, rd, b, label = H.make_rays(n, 31)
    blk = G.make_block(c, max_rays=128, dtype=dtype)
    blk.set_params(H.make_params(c, 11))
    sig_t, craw_t, sig_i, craw_i = _stage(blk, c, ro, rd, b, label)
    blk.close()
    S = sig_t.shape[1]
    for name, a, r in (("sigma", sig_i[:, :S - 1], sig_t[:, :S - 1]), ("craw", craw_i, craw_t)):   # the last sample has no density
        scale = np.abs(r).max()
        d = np.abs(a - r) / scale
        if dtype == "f16x3":
            assert d.max() <= 1e-5, (name, d.max())
        else:
            assert d.max() <= 2e-2 and d.mean() <= 1e-4, (name, d.max(), d.mean())


@pytest.mark.parametrize("dtype", ["f16x3", "bf16"])
def test_inference_streams_follow_the_optimizer(dtype):
    """render, one training step, render again: the second render must be the new parameters' (the 16x16x32-ordered streams
    are re-packed with the parameters -- on demand for f16x3), i.e. equal to a fresh handle loaded with them"""
    c = dict(H.CONFIGS["small_dir"])
    n = 48
    ro, rd, b, label = H.make_rays(n, 5)
    a = G.make_block(c, max_rays=64, dtype=dtype)
    a.set_params(H.make_params(c, 2))
    before = a.evaluate(ro, rd, b, rd)
    for _ in range(3):
        a.train_step(ro, rd, b, rd, label)
    after = a.evaluate(ro, rd, b, rd)
    fresh = G.make_block(c, max_rays=64, dtype=dtype)
    fresh.set_params(a.get_params())
    ref = fresh.evaluate(ro, rd, b, rd)
    a.close(); fresh.close()
    np.testing.assert_array_equal(after, ref)
    assert np.abs(after - before).max() > 1e-4
