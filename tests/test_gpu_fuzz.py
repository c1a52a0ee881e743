"""Seeded random configurations: network shape (depth, width, skips, encoding lengths, colour head), sampling options
(hierarchical or not, sample counts that are no multiple of anything, linDisp, perturb, raw noise, background) and batch
sizes drawn at random, rendered and differentiated on the GPU and by the oracle with the same explicit random tensors.
The hand-picked cases elsewhere cover the shapes someone thought of; this covers the combinations nobody did.
Bars: sampler outputs bit-exact on identical inputs, render 2e-4 abs and gradients 1e-3 (whole) / 5e-3 (per tensor) with
the oracle evaluated at the device's fine depths (tests/test_gpu_train.py: test_gradients_full explains why).
The seeds kept here are green; of 2 500 explored (scripts/fuzz_explore.py, profiles/r3_fuzz_*.txt) fourteen miss a gradient bar
(355, 715, 1236, 1237, 1460, 1572, 1631, 1644, 1883, 1958, 1972, 1974, 2092, 2476), each by ONE ReLU whose pre-activation is within
fp32 rounding of zero at one sample (1e-8 ... 7e-7 in magnitude against a median of 0.2-1.6; 355: -4.2e-8), where device and oracle
legitimately take different sides: one row of one matrix carries that tensor's whole error, the layers above it agree to 1e-7, the
layers below it move by 0.2-1.7 % of their norms, and in ten of the fourteen the other device arithmetic (f32 / f16x3) agrees
with the oracle on the same seed (scripts/fuzz_one.py prints where the error sits and the unit's pre-activations in float64;
profiles/r3_fuzz_fence_cases.txt)."""
import numpy as np
import pytest

import gpu_util as G
import helpers as H
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def draw(seed):
    r = np.random.RandomState(1000 + seed)
    D = int(r.randint(1, 9))
    W = int(r.choice([32, 64, 128, 256], p=[0.2, 0.4, 0.25, 0.15]))
    skips = sorted(int(x) for x in r.choice(np.arange(0, max(D - 1, 1)), size=int(r.randint(0, min(3, max(D - 1, 1)) + 1)), replace=False)) if D >= 2 else []
    mode = int(r.randint(0, 3))
    hier = bool(r.randint(0, 2))
    N = int(r.randint(3, 41))
    c = H.cfg(D=D, W=W, skips=skips, posL=int(r.randint(0, 11)), dirL=int(r.randint(0, 5)), useDir=mode != 0, useSH=mode == 2,
              useHierarchical=hier, NSamples=N, NImportance=int(r.randint(1, 51)) if hier else 0, whiteBkgd=bool(r.randint(0, 2)),
              linDisp=bool(r.randint(0, 2)), perturb=bool(r.randint(0, 2)), rawNoiseStd=float(r.choice([0.0, 1.0])))
    return c, int(r.randint(1, 71)), r


@pytest.mark.parametrize("seed", range(48))
def test_random_configuration(seed):
    c, n, r = draw(seed)
    N, NI = c["NSamples"], c["NImportance"]
    S = N + NI if c["useHierarchical"] else N
    ro, rd, b, label = H.make_rays(n, 200 + seed)
    params = H.make_params(c, 50 + seed)
    rand = {}
    if c["perturb"]:
        rand["t_rand"] = H.det_uniform(n * N, seed + 7).reshape(n, N)
        if c["useHierarchical"]:
            rand["u_rand"] = H.det_uniform(n * NI, seed + 8).reshape(n, NI)
    if c["rawNoiseStd"] > 0:
        rand["noise_c"] = H.det_normal(n * N, seed + 9).reshape(n, N)
        rand["noise_f"] = H.det_normal(n * S, seed + 10).reshape(n, S)
    o = Oracle(c)
    for dtype in (("f32", "f16x3") if c["W"] >= 64 else ("f32",)):
        blk = G.make_block(c, max_rays=max(n, 8), dtype=dtype)
        try:
            blk.set_params(params)
            # eval render (no noise in eval mode, nerfBlock.scala:21; the perturb tensors apply)
            ev = {k: v for k, v in rand.items() if not k.startswith("noise")}
            fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, rand=ev, debug=True)
            ref = o.forward(params, ro, rd, b, rd, training=False, rand=dict(ev, z_fine_in=dbg["z_fine"]))
            np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"], err_msg=f"{c} {dtype}")
            assert (np.diff(dbg["z_fine"], axis=1) >= 0).all()
            np.testing.assert_allclose(fine, ref["rgb_fine"], rtol=0, atol=2e-4, err_msg=f"{c} {dtype}")
            if c["useHierarchical"]:   # the sampler on the DEVICE's own coarse weights: bit-exact
                cdf, ind, zs, zf = o.sample_pdf(dbg["w_coarse"], dbg["z_coarse"], u_rand=ev.get("u_rand"))
                np.testing.assert_array_equal(dbg["z_fine"], zf, err_msg=f"{c} {dtype}")
            # training pass
            lf, lc, dbg = blk.train_backward(ro, rd, b, rd, label, rand=rand, debug=True)
            tr = o.train_grads(params, ro, rd, b, rd, label, rand=dict(rand, z_fine_in=dbg["z_fine"]))
            assert lf == pytest.approx(tr["loss_fine"], rel=1e-4, abs=1e-7), (c, dtype)
            if c["useHierarchical"]:
                assert lc == pytest.approx(tr["loss_coarse"], rel=1e-4, abs=1e-7), (c, dtype)
            H.assert_grads_close(c, blk.get_grads(), tr["grads"], whole=1e-3)
        finally:
            blk.close()
