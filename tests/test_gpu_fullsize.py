"""Size-independent properties at BASELINE.json's full sizes (the oracle would take minutes there): rays are
independent in the forward pass (nerfBlock.scala has no cross-ray op) and couple in training only through the
mean loss, so
  * a 400x400 frame rendered in one call, in two halves, or in permuted ray order gives bit-identical pixels;
  * the gradient of a 4096-ray batch is the mean of the gradients of its two halves -- exactly the identity the
    data-parallel step relies on (g GPUs x B rays == one device with g*B rays).
Both dtypes, through the C ABI."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_util
    return gpu_util


def _frame(G, blk, hw=400):
    import math
    n = hw * hw
    ro, rd, bd = G.empty(n, 3), G.empty(n, 3), G.empty(n, 2)
    cam = 4.0 * np.array([math.cos(0.7) * math.cos(0.5), math.sin(0.7) * math.cos(0.5), math.sin(0.5)])
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.stack([right, up, -fwd, cam], 1).astype(np.float32)
    blk.k_gen_rays(hw, hw, 0.5 * hw / math.tan(0.5 * 0.6911112070083618), c2w, 2.0, 6.0, False, ro, rd, bd)
    blk.synchronize()
    return ro, rd, bd


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16"])
def test_frame_split_and_permutation_invariance(G, dtype):
    import torch
    c = H.CONFIGS["full"]
    n = 160000
    blk = G.make_block(c, max_rays=n, dtype=dtype)
    blk.init_params(3)
    ro, rd, bd = _frame(G, blk)
    whole, parts, perm = G.empty(n, 3), G.empty(n, 3), G.empty(n, 3)
    blk.render_dev(ro, rd, bd, rd, n, whole)
    h = 77777                                     # ragged split: neither part is a multiple of any tile size
    blk.render_dev(ro[:h], rd[:h], bd[:h], rd[:h], h, parts[:h])
    blk.render_dev(ro[h:], rd[h:], bd[h:], rd[h:], n - h, parts[h:])
    p = torch.randperm(n, generator=torch.Generator().manual_seed(1)).cuda()
    blk.render_dev(ro[p].contiguous(), rd[p].contiguous(), bd[p].contiguous(), rd[p].contiguous(), n, perm)
    blk.synchronize()
    whole = whole.cpu().numpy()
    assert np.isfinite(whole).all() and whole.std() > 1e-3
    np.testing.assert_array_equal(parts.cpu().numpy(), whole)
    np.testing.assert_array_equal(perm.cpu().numpy(), whole[p.cpu().numpy()])
    blk.close()


# f16x3: the weight-gradient operand scale comes from the batch's largest gradient, so shards and whole batch round differently
@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("f16x3", 5e-5), ("bf16", 2e-5)])
def test_batch_gradient_is_mean_of_shard_gradients(G, dtype, tol):
    c = H.CONFIGS["full"]
    B = 4096
    blk = G.make_block(c, max_rays=B, dtype=dtype)
    blk.init_params(5)
    ro, rd, b, label = H.make_rays(B, 4)
    lf, lc = blk.train_backward(ro, rd, b, rd, label)
    g_all = blk.get_grads().astype(np.float64)
    acc = np.zeros_like(g_all)
    losses = []
    for s in (slice(0, B // 2), slice(B // 2, B)):
        l = blk.train_backward(ro[s], rd[s], b[s], rd[s], label[s])
        acc += blk.get_grads()
        losses.append(l)
    acc *= 0.5
    # identical per-sample arithmetic; only the fp32 summation order over samples differs (atomics, slice cuts)
    assert np.linalg.norm(acc - g_all) <= tol * np.linalg.norm(g_all)
    assert abs(0.5 * (losses[0][0] + losses[1][0]) - lf) <= 1e-5 * max(1.0, abs(lf))
    assert abs(0.5 * (losses[0][1] + losses[1][1]) - lc) <= 1e-5 * max(1.0, abs(lc))
    blk.close()
