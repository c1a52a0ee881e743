"""The library-generated random branches: givePerturb / givePerturbPDF / addNoise with NO explicit
random tensors (nerfBlock.scala:55,71,21 draw manager.randomUniform / randomNormal there) and nerfhip_init_params
(XavierInitializer(GAUSSIAN, IN, 2) = N(0, 2/fan_in), biases 0).

The reference's random numbers themselves cannot be reproduced (libtorch's generator, never seeded in package nerf),
so parity here means: (1) the implicit branch of every kernel equals, BIT FOR BIT, the explicit-tensor branch fed with
the values of the library's own Philox stream (nerfhip_k_random) -- and the explicit branch is oracle-checked in
test_gpu_render / test_gpu_train; (2) the streams have the distribution the reference draws from (moments within
stated bounds); (3) determinism in (seed, step), change across steps, seeds and streams."""
import numpy as np
import pytest

import helpers as H
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

PERTURB, PDF, NOISE_C, NOISE_F, INIT = 1, 2, 3, 4, 5


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available()
    import gpu_util
    return gpu_util


def _stream(G, blk, stream, seed, step, n):
    out = G.empty(n)
    blk.k_random(stream, seed, step, n, out)
    blk.synchronize()
    return out.cpu().numpy()


def test_streams_moments_and_independence(G):
    """U(0,1): mean 1/2 +- 2e-3, variance 1/12 +- 1e-3, never 0 or 1; N(0,1): mean 0 +- 5e-3, variance 1 +- 1e-2,
    |skew| < 2e-2, kurtosis 3 +- 5e-2 (n = 2^20; 4-5 sigma bounds).  Distinct (stream | step | seed) -> distinct and
    uncorrelated values; equal keys -> identical values."""
    blk = G.make_block(H.CONFIGS["small_opts"])
    n = 1 << 20
    u = _stream(G, blk, PERTURB, 7, 0, n)
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    g = _stream(G, blk, NOISE_F, 7, 0, n).astype(np.float64)
    assert np.isfinite(g).all()
    assert abs(g.mean()) < 5e-3 and abs(g.var() - 1) < 1e-2
    assert abs((g ** 3).mean()) < 2e-2 and abs((g ** 4).mean() - 3) < 5e-2
    np.testing.assert_array_equal(u, _stream(G, blk, PERTURB, 7, 0, n))           # stateless: same key, same values
    for other in (_stream(G, blk, PDF, 7, 0, n), _stream(G, blk, PERTURB, 7, 1, n), _stream(G, blk, PERTURB, 8, 0, n)):
        assert (other != u).mean() > 0.999
        assert abs(np.corrcoef(u, other)[0, 1]) < 5e-3
    # neighbouring indices are uncorrelated too
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5e-3
    blk.close()


def test_perturb_implicit_equals_explicit_and_oracle(G):
    """k_stratified with t_rand = NULL == k_stratified with the Philox values passed in == the oracle on those values;
    every depth stays inside its bin [mid_{i-1}, mid_i] and the row ascends; the next call uses the next step."""
    c = H.CONFIGS["small_opts"]          # perturb on, linDisp on
    n, N = 500, c["NSamples"]
    _, _, bounds, _ = H.make_rays(n, 3)
    blk = G.make_block(c, seed=99)
    o = Oracle(c)
    db = G.dev(bounds)
    z1, z2, ze = G.empty(n, N), G.empty(n, N), G.empty(n, N)
    _, step = blk.rng_step()
    blk.k_stratified(db, None, n, z1)
    assert blk.rng_step()[1] == step + 1
    tr = G.empty(n * N)
    blk.k_random(PERTURB, 99, step, n * N, tr)
    blk.k_stratified(db, tr, n, ze)
    blk.k_stratified(db, None, n, z2)     # one step later
    blk.synchronize()
    z1, z2, ze, tr = z1.cpu().numpy(), z2.cpu().numpy(), ze.cpu().numpy(), tr.cpu().numpy().reshape(n, N)
    np.testing.assert_array_equal(z1, ze)
    np.testing.assert_array_equal(z1, o.stratified(bounds, tr))
    assert (z1 != z2).mean() > 0.99
    z0 = Oracle(dict(c, perturb=False)).stratified(bounds)
    mid = 0.5 * (z0[:, 1:] + z0[:, :-1])
    lower = np.concatenate([z0[:, :1], mid], 1)
    upper = np.concatenate([mid, z0[:, -1:]], 1)
    assert (z1 >= lower).all() and (z1 <= upper).all()
    assert (np.diff(z1, axis=1) >= 0).all()
    # a second handle with the same seed replays the sequence; another seed does not
    b2, b3 = G.make_block(c, seed=99), G.make_block(c, seed=100)
    y2, y3 = G.empty(n, N), G.empty(n, N)
    b2.k_stratified(db, None, n, y2); b3.k_stratified(db, None, n, y3)
    b2.synchronize(); b3.synchronize()
    np.testing.assert_array_equal(y2.cpu().numpy(), z1)
    assert (y3.cpu().numpy() != z1).mean() > 0.99
    for b in (blk, b2, b3):
        b.close()


def test_pdf_implicit_equals_explicit_and_oracle(G):
    """k_sample_pdf with u_rand = NULL (perturb on) == the explicit branch on the Philox values == the oracle on them
    (cdf, bin indices, samples, merged depths bit-exact); u is sorted per ray (bin indices ascend)."""
    c = H.CONFIGS["small_opts"]
    n, N, NI = 300, c["NSamples"], c["NImportance"]
    w = H.det_uniform(n * N, 5).reshape(n, N) ** 3
    z = np.sort(H.det_uniform(n * N, 6, 2.0, 6.0).reshape(n, N), axis=1)
    blk = G.make_block(c, seed=5)
    o = Oracle(c)
    outs = []
    _, step = blk.rng_step()
    ur = G.empty(n * NI)
    blk.k_random(PDF, 5, step, n * NI, ur)
    for u_in in (None, ur):
        cdf, ind, zs, zf = G.empty(n, N - 1), G.empty(n, NI, dtype=__import__("torch").int32), G.empty(n, NI), G.empty(n, N + NI)
        blk.k_sample_pdf(G.dev(w), G.dev(z), u_in, n, cdf, ind, zs, zf)
        blk.synchronize()
        outs.append([t.cpu().numpy() for t in (cdf, ind, zs, zf)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    ref = o.sample_pdf(w, z, ur.cpu().numpy().reshape(n, NI))
    for a, r in zip(outs[0], ref):
        np.testing.assert_array_equal(a, r)
    assert (np.diff(outs[0][1], axis=1) >= 0).all() and (np.diff(outs[0][3], axis=1) >= 0).all()
    blk.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_forward_implicit_randoms_equal_explicit(G, dtype):
    """The whole stochastic training step (perturb + PDF perturb + raw-density noise on both networks, all drawn by the
    library) == the same step with the four Philox tensors passed in explicitly: outputs, losses and the gradient
    (backward re-draws the noise of the step the forward used).  fp32 gradients differ only by atomics order."""
    c = H.CONFIGS["small_opts"]
    n, N, NI = 96, c["NSamples"], c["NImportance"]
    S = N + NI
    ro, rd, b, label = H.make_rays(n, 4)
    params = H.make_params(c, 2)
    a = G.make_block(c, seed=11, dtype=dtype)
    e = G.make_block(c, seed=12345, dtype=dtype)    # another seed: it must not matter when everything is explicit
    for blk in (a, e):
        blk.set_params(params)
    for it in range(2):                              # two steps: the streams advance
        _, step = a.rng_step()
        rand = {"t_rand": _stream(G, a, PERTURB, 11, step, n * N).reshape(n, N),
                "u_rand": _stream(G, a, PDF, 11, step, n * NI).reshape(n, NI),
                "noise_c": _stream(G, a, NOISE_C, 11, step, n * N).reshape(n, N),
                "noise_f": _stream(G, a, NOISE_F, 11, step, n * S).reshape(n, S)}
        la = a.train_backward(ro, rd, b, rd, label, debug=True)
        assert a.rng_step()[0] == step
        le = e.train_backward(ro, rd, b, rd, label, rand=rand, debug=True)
        for k in ("z_coarse", "z_fine", "w_coarse", "w_fine", "sigma_f", "craw_f"):
            np.testing.assert_array_equal(la[2][k], le[2][k], err_msg=k)
        assert la[:2] == le[:2]
        ga, ge = a.get_grads(), e.get_grads()
        np.testing.assert_allclose(ga, ge, rtol=0, atol=2e-6 * np.abs(ge).max())
        if it == 0:
            first = la[2]["z_fine"].copy()
            if dtype == "f32":   # and the explicit branch is the oracle's
                ref = Oracle(c).train_grads(params, ro, rd, b, rd, label, rand=rand)
                assert la[0] == pytest.approx(ref["loss_fine"], rel=1e-4)
                assert np.linalg.norm(ga - ref["grads"]) / np.linalg.norm(ref["grads"]) < 1e-3
    assert (la[2]["z_fine"] != first).mean() > 0.5   # a new step draws new numbers
    a.close(); e.close()


def test_eval_draws_no_noise_but_perturbs(G):
    """training=false: addNoise is off (nerfBlock.scala:21) while perturb stays on (:50-57 has no training switch)."""
    c = H.CONFIGS["small_opts"]
    n, N, NI = 64, c["NSamples"], c["NImportance"]
    ro, rd, b, _ = H.make_rays(n, 6)
    params = H.make_params(c, 2)
    blk = G.make_block(c, seed=3)
    blk.set_params(params)
    _, step = blk.rng_step()
    rand = {"t_rand": _stream(G, blk, PERTURB, 3, step, n * N).reshape(n, N),
            "u_rand": _stream(G, blk, PDF, 3, step, n * NI).reshape(n, NI)}
    fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, debug=True)
    ref = Oracle(c).forward(params, ro, rd, b, rd, training=False, rand=rand)
    np.testing.assert_array_equal(dbg["z_coarse"], ref["z_coarse"])
    assert np.abs(fine - ref["rgb_fine"]).max() < 1e-3 and (np.abs(fine - ref["rgb_fine"]) <= 2e-4).mean() > 0.97
    blk.close()


@pytest.mark.parametrize("name", ["full", "shipped", "small_nodir"])
def test_init_params_distribution(G, name):
    """nerfhip_init_params: every matrix ~ N(0, 2/fan_in) -- sample std within 2 % (matrices of >= 8192 entries; 5 sigma
    of the estimator otherwise), |mean| < 5 sd/sqrt(n) -- every bias exactly 0, Adam state cleared, deterministic in the
    seed, and equal to sd * (INIT stream at the flat parameter index)."""
    c = H.CONFIGS[name]
    blk = G.make_block(c, max_rays=64)
    blk.init_params(42)
    p = blk.get_params()
    assert p.size == H.param_count(c)
    nets = 2 if c["useHierarchical"] else 1
    raw = _stream(G, blk, INIT, 42, 0, p.size)
    off = 0
    for net in range(nets):
        for nm, r, k in H.layout(c):
            seg = p[off:off + r * k]
            if nm.startswith("w"):
                sd = np.float32(np.sqrt(2.0 / k))
                cnt = r * k
                tol = 0.02 if cnt >= 8192 else 5.0 / np.sqrt(2.0 * cnt)
                assert abs(seg.std() / sd - 1) < tol, (net, nm, seg.std(), sd)
                assert abs(seg.mean()) < 5 * sd / np.sqrt(cnt), (net, nm)
                np.testing.assert_array_equal(seg, raw[off:off + cnt] * sd)
            else:
                assert (seg == 0).all(), (net, nm)
            off += r * k
    if nets == 2:   # the two networks are different draws
        half = p.size // 2
        assert (p[:half] != p[half:]).mean() > 0.9
    m, v, step = blk.get_opt_state()
    assert step == 0 and not m.any() and not v.any()
    # after training steps init_params resets the optimizer too
    ro, rd, b, label = H.make_rays(32, 1)
    blk.train_step(ro, rd, b, rd, label)
    assert blk.get_opt_state()[2] == 1
    blk.init_params(42)
    np.testing.assert_array_equal(blk.get_params(), p)
    m, v, step = blk.get_opt_state()
    assert step == 0 and not m.any() and not v.any()
    blk.init_params(43)
    assert (blk.get_params() != p).mean() > 0.9
    blk.close()
