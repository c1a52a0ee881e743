"""Host side of LLFF scenes (nerf_amd/llff.py <- nerf/llff.scala, llff branch of nerf/getDataSet.scala).  No scene and
no fixtures ship with the reference (parity unpinned), so: closed-form / invariant checks on a synthetic
forward-facing scene written here, and the device rays against the oracle's ray generator."""
import math
import os

import numpy as np
import pytest

import helpers as H
from nerf_amd import llff as L


def _rot(ax, ay):
    cx, sx, cy, sy = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay)
    return np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])


def _make_scene(root, n=9, hw=(12, 16), factor=2, with_reduced=True):
    from PIL import Image
    rng = np.random.default_rng(1)
    arr = np.zeros((n, 17))
    for i in range(n):
        R = _rot(0.1 * rng.standard_normal(), 0.1 * rng.standard_normal())     # columns = camera x, y, z in LLFF's axis order
        R_llff = np.stack([-R[:, 1], R[:, 0], R[:, 2]], 1)                    # stored as [-y, x, z] ("down, right, back")
        t = np.array([rng.uniform(-1, 1), rng.uniform(-.5, .5), rng.uniform(-.2, .2)]) + [5.0, -3.0, 2.0]
        p = np.concatenate([R_llff, t[:, None], np.array([[hw[0] * factor], [hw[1] * factor], [40.0]])], 1)
        arr[i, :15] = p.reshape(-1)
        arr[i, 15:] = [rng.uniform(2.0, 3.0), rng.uniform(20.0, 30.0)]
    np.save(os.path.join(root, "posesBounds.npy"), arr)
    os.makedirs(os.path.join(root, "images"))
    full = []
    for i in range(n):
        px = rng.integers(0, 256, (hw[0] * factor, hw[1] * factor, 3), dtype=np.uint8)
        Image.fromarray(px, "RGB").save(os.path.join(root, "images", f"img_{i:03d}.png"))
        full.append(px)
    if with_reduced:
        os.makedirs(os.path.join(root, f"images_{factor}"))
        for i in range(n):
            red = full[i].reshape(hw[0], factor, hw[1], factor, 3).mean((1, 3)).astype(np.uint8)
            Image.fromarray(red, "RGB").save(os.path.join(root, f"images_{factor}", f"img_{i:03d}.png"))
    return arr, full


def test_small_linear_algebra():
    x, y = np.array([1.0, 2.0, -3.0]), np.array([0.5, -1.0, 4.0])
    np.testing.assert_allclose(L.cross(x, y), np.cross(x, y))
    np.testing.assert_allclose(np.linalg.norm(L.normalize(x)), 1.0)
    rng = np.random.default_rng(0)
    for _ in range(20):
        m = rng.standard_normal((3, 3)) + 2 * np.eye(3)
        np.testing.assert_allclose(L.inv(m) @ m, np.eye(3), atol=1e-9)


def test_recenter_invariants():
    rng = np.random.default_rng(2)
    poses = np.stack([np.concatenate([_rot(*(0.2 * rng.standard_normal(2))), rng.standard_normal((3, 1)) + 4], 1) for _ in range(7)], 0)
    out = L.recenter(poses.copy())
    np.testing.assert_allclose(out[..., 3].mean(0), 0, atol=1e-12)                   # positions centred
    zsum = out[..., 2].sum(0)
    np.testing.assert_allclose(zsum / np.linalg.norm(zsum), [0, 0, 1], atol=1e-9)      # mean viewing axis = world z
    assert abs(out[..., 1].sum(0)[0]) < 1e-9                                           # summed y axis has no x component
    for p in out:                                                                      # still rotations
        np.testing.assert_allclose(p[:, :3].T @ p[:, :3], np.eye(3), atol=1e-9)


def test_spiral_render_path():
    rads = np.array([0.9, 0.45, 0.18])
    rp = L.get_render_poses(rads, 7.5, np.float32(.5), 2, 120)
    assert rp.shape == (120, 3, 4)
    c = rp[..., 3]
    assert (np.abs(c) <= rads + 1e-12).all()
    np.testing.assert_allclose(c[0], [0.9, 0, 0], atol=1e-12)                          # theta = 0
    np.testing.assert_allclose(c[60], c[0], atol=1e-5)                                 # one full turn after N / rots poses
    for p in rp[::17]:
        g = p[:, :3].T @ p[:, :3]
        np.testing.assert_allclose(g - np.diag(np.diag(g)), 0, atol=1e-9)              # orthogonal axes, z of unit length;
        assert g[2, 2] == pytest.approx(1.0) and g[0, 0] <= 1.0 + 1e-12                # x = cross(up, z) is NOT normalised
        assert g[0, 0] == pytest.approx(g[1, 1])                                       # in the reference (llff.scala:151-152)
        look = np.array([0, 0, -7.5]) - p[:, 3]
        np.testing.assert_allclose(-p[:, 2], look / np.linalg.norm(look), atol=1e-9)   # camera looks down its -z at the focal point


@pytest.mark.parametrize("with_reduced", [True, False])
def test_load_llff_data(tmp_path, with_reduced):
    arr, full = _make_scene(str(tmp_path), with_reduced=with_reduced)
    poses, render, images, bds = L.load_llff_data(str(tmp_path), factor=2, bd_factor=.75)
    n = arr.shape[0]
    assert poses.shape == (n, 3, 5) and render.shape == (120, 3, 4) and images.shape == (n, 12, 16, 3)
    assert poses.dtype == np.float32 and render.dtype == np.float32 and bds.dtype == np.float32
    np.testing.assert_array_equal(poses[:, 0, 4], 12); np.testing.assert_array_equal(poses[:, 1, 4], 16)
    np.testing.assert_allclose(poses[:, 2, 4], 20.0)                                   # focal / factor
    assert bds.min() == pytest.approx(1 / .75, rel=1e-6)                                # scaled so that min(bds) * bdFactor = 1
    np.testing.assert_allclose(bds, arr[:, 15:] / (arr[:, 15:].min() * .75), rtol=1e-6)
    np.testing.assert_allclose(poses[:, :, 3].mean(0), 0, atol=1e-6)
    box = full[3].reshape(12, 2, 16, 2, 3).mean((1, 3))
    if with_reduced:                                                                   # the stored reduced PNG (uint8) is used as is
        np.testing.assert_array_equal(images[3], box.astype(np.uint8).astype(np.float32) / np.float32(255))
    else:                                                                              # reduced on the fly: fp32 box mean / 255
        np.testing.assert_allclose(images[3], box / 255, atol=1e-6)
    # axis swap [y, -x, z] (llff.scala:221) makes the stored [-y, x, z] columns a proper camera frame again
    for p in poses:
        assert np.linalg.det(p[:, :3].astype(np.float64)) == pytest.approx(1.0, abs=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ndc", [True, False])
def test_llff_rays_on_device(tmp_path, ndc):
    import gpu_util as G
    from oracle.oracle import Oracle
    _make_scene(str(tmp_path))
    c = H.CONFIGS["small_dir"]
    blk = G.make_block(c)
    blk.init_params(1)
    ds = L.LlffRays(blk, str(tmp_path), factor=2, llff_hold=4, ndc=ndc)
    assert ds.i_test == [0, 4, 8] and ds.i_train == [1, 2, 3, 5, 6, 7]
    assert (ds.H, ds.W) == (12, 16) and ds.focal == pytest.approx(20.0)
    o = Oracle(c)
    n = ds.H * ds.W
    ro, rd, b, lab = ds.splits["train"]
    assert ro.shape == (6 * n, 3) and lab.shape == (6 * n, 3)
    for k, i in enumerate(ds.i_train[:3]):
        eo, ed, eb = o.gen_rays(ds.H, ds.W, ds.focal, ds.poses[i], ds.near, ds.far, ndc)
        np.testing.assert_array_equal(ro[k * n:(k + 1) * n].cpu().numpy(), eo)
        np.testing.assert_array_equal(rd[k * n:(k + 1) * n].cpu().numpy(), ed)
        np.testing.assert_array_equal(b[k * n:(k + 1) * n].cpu().numpy(), eb)
    if ndc:
        assert float(b[:, 0].abs().max()) == 0.0                                       # bounds (0, |d'|), getDataSet.scala:112
    (bo, bd, bb), bl = next(iter(ds.batches("train", 128, seed=0)))
    lf = np.zeros(2, np.float32)
    blk.train_step_dev(bo.contiguous(), bd.contiguous(), bb.contiguous(), bd.contiguous(), bl.contiguous(), 128, lf)
    blk.synchronize()
    assert np.isfinite(lf).all()
    blk.close()
