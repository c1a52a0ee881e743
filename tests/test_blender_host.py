"""Host side of Blender-format scenes (nerf_amd/blender.py <- nerf/blender.scala, nerf/getDataSet.scala,
nerf/nerfDataSet.scala).  The reference ships no scene and no fixtures (parity unpinned), so
these are closed-form checks on a tiny synthetic scene directory written here; the ray generator itself is pinned
bit-exactly against the oracle (tests/test_gpu_render.py::test_gen_rays_bit_exact) and again below per pose."""
import json
import math
import os

import numpy as np
import pytest

import helpers as H
from nerf_amd import blender as B

ANGLE = 0.6911112070083618


def _make_scene(root, n=(3, 2, 4), hw=(6, 8)):
    from PIL import Image
    rng = np.random.default_rng(0)
    imgs = {}
    for s, k in zip(("train", "val", "test"), n):
        frames = []
        os.makedirs(os.path.join(root, s), exist_ok=True)
        for i in range(k):
            px = rng.integers(0, 256, (hw[0], hw[1], 4), dtype=np.uint8)
            Image.fromarray(px, "RGBA").save(os.path.join(root, s, f"r_{i}.png"))
            pose = B.get_render_poses(0.3 * i + 0.1, 0.4, 4.0)
            m = np.vstack([pose, [[0, 0, 0, 1]]]).astype(float)
            frames.append({"file_path": f"./{s}/r_{i}", "rotation": 0.012, "transform_matrix": m.tolist()})
            imgs[(s, i)] = (px, pose)
        with open(os.path.join(root, f"transforms_{s}.json"), "w") as f:
            json.dump({"camera_angle_x": ANGLE, "frames": frames}, f)
    return imgs


def test_render_pose_closed_form():
    # theta = phi = 0: the "origin" matrix itself (camera at (r,0,0) looking at the origin, blender.scala:53-57)
    np.testing.assert_array_equal(B.get_render_poses(0.0, 0.0, 4.0),
                                  np.array([[0, 0, 1, 4], [1, 0, 0, 0], [0, 1, 0, 0]], np.float32))
    for th, ph in ((0.7, math.pi / 6), (2.9, 0.2), (5.5, -0.4)):
        p = B.get_render_poses(th, ph, 4.0).astype(np.float64)
        R, t = p[:, :3], p[:, 3]
        np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-6)                      # a rotation
        np.testing.assert_allclose(t, 4.0 * np.array([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph),
                                                      math.sin(ph)]), atol=1e-5)
        np.testing.assert_allclose(R[:, 2], t / 4.0, atol=1e-6)                         # looks along -z at the origin
        assert abs(R[2, 0]) < 1e-6                                                      # camera x axis in the xy plane


def test_load_blender_data(tmp_path):
    imgs = _make_scene(str(tmp_path))
    poses, render, images, (Hh, Ww, focal), i_split = B.load_blender_data(str(tmp_path), half_res=False, test_skip=2)
    assert i_split == [3, 4, 6]                        # train: all 3; val: 1 of 2; test: 2 of 4 (testSkip = 2)
    assert (Hh, Ww) == (6, 8) and images.shape == (6, 6, 8, 4) and poses.shape == (6, 3, 4)
    assert focal == pytest.approx(0.5 * 8 / math.tan(0.5 * ANGLE), rel=1e-6)
    np.testing.assert_array_equal(images[0], imgs[("train", 0)][0].astype(np.float32) / np.float32(255))
    np.testing.assert_array_equal(images[4], imgs[("test", 0)][0].astype(np.float32) / np.float32(255))
    np.testing.assert_array_equal(images[5], imgs[("test", 2)][0].astype(np.float32) / np.float32(255))
    np.testing.assert_array_equal(poses[3], imgs[("val", 0)][1])
    # 40 render poses: radius 4, phi = pi/6, theta = 2 pi i / 40 (blender.scala:145)
    assert render.shape == (40, 3, 4)
    np.testing.assert_allclose(np.linalg.norm(render[:, :, 3], axis=1), 4.0, atol=1e-5)
    np.testing.assert_allclose(render[:, 2, 3], 4.0 * math.sin(math.pi / 6), atol=1e-5)
    np.testing.assert_allclose(np.arctan2(render[7, 1, 3], render[7, 0, 3]), 2 * math.pi * 7 / 40, atol=1e-5)
    # testSkip == 0 means "no skipping" (blender.scala:123)
    assert B.load_blender_data(str(tmp_path), test_skip=0)[4] == [3, 5, 9]


def test_half_res_is_the_2x2_box_mean(tmp_path):
    imgs = _make_scene(str(tmp_path))
    _, _, images, (Hh, Ww, focal), _ = B.load_blender_data(str(tmp_path), half_res=True, test_skip=1)
    assert (Hh, Ww) == (3, 4) and images.shape[1:] == (3, 4, 4)
    assert focal == pytest.approx(0.5 * 4 / math.tan(0.5 * ANGLE), rel=1e-6)
    px = imgs[("train", 1)][0].astype(np.int64)
    box = (px[0::2, 0::2] + px[0::2, 1::2] + px[1::2, 0::2] + px[1::2, 1::2]) // 4      # truncating mean
    np.testing.assert_array_equal(images[1], box.astype(np.float32) / np.float32(255))


def test_ray_batches_cover_every_ray_once():
    import torch
    n = 1000
    data = (torch.arange(n).float().unsqueeze(1).repeat(1, 3), torch.arange(n).float().unsqueeze(1).repeat(1, 3),
            torch.arange(n).float().unsqueeze(1).repeat(1, 2))
    lab = torch.arange(n).float().unsqueeze(1).repeat(1, 3)
    rb = B.RayBatches(data, lab, 96, seed=3)
    assert len(rb) == 11                               # ceil(1000 / 96), the last batch short (nerfDataSet.scala:17,58)
    seen, sizes = [], []
    for (o, d, b), l in rb:
        assert torch.equal(o[:, 0], l[:, 0]) and torch.equal(d[:, 0], b[:, 0])   # data and label stay aligned
        seen.append(o[:, 0]); sizes.append(o.shape[0])
    assert sizes == [96] * 10 + [40]
    assert sorted(torch.cat(seen).tolist()) == list(range(n))
    first = torch.cat(seen)
    second = torch.cat([o[:, 0] for (o, _, _), _ in rb])
    assert not torch.equal(first, second)              # a fresh permutation every epoch


@pytest.mark.gpu
def test_blender_rays_on_device(tmp_path):
    """getDataSet.apply (blender): rays of every pose == the oracle's ray generator bit for bit, labels == the
    white-background blend, splits by iSplit; and a scene batch trains."""
    import torch
    import gpu_util as G
    from oracle.oracle import Oracle
    imgs = _make_scene(str(tmp_path))
    c = H.CONFIGS["small_dir"]
    blk = G.make_block(c)
    blk.init_params(1)
    ds = B.BlenderRays(blk, str(tmp_path), half_res=False, test_skip=1, white_bkgd=True)
    o = Oracle(c)
    n = ds.H * ds.W
    ro, rd, b, lab = ds.splits["train"]
    assert ro.shape == (3 * n, 3) and lab.shape == (3 * n, 3)
    assert ds.splits["val"][0].shape[0] == 2 * n and ds.splits["test"][0].shape[0] == 4 * n
    for i in range(3):
        eo, ed, eb = o.gen_rays(ds.H, ds.W, ds.focal, imgs[("train", i)][1], 2.0, 6.0, False)
        np.testing.assert_array_equal(ro[i * n:(i + 1) * n].cpu().numpy(), eo)
        np.testing.assert_array_equal(rd[i * n:(i + 1) * n].cpu().numpy(), ed)
        np.testing.assert_array_equal(b[i * n:(i + 1) * n].cpu().numpy(), eb)
        px = imgs[("train", i)][0].astype(np.float32) / np.float32(255)
        want = px[..., :3] * px[..., 3:] + (-(px[..., 3:] - np.float32(1)))
        np.testing.assert_array_equal(lab[i * n:(i + 1) * n].cpu().numpy(), want.reshape(-1, 3))
    eo, ed, eb = o.gen_rays(ds.H, ds.W, ds.focal, ds.render_poses[5], 2.0, 6.0, False)
    np.testing.assert_array_equal(ds.render[1][5 * n:6 * n].cpu().numpy(), ed)
    losses = []
    for (bo, bd, bb), bl in ds.batches("train", 64, seed=0):
        lf = np.zeros(2, np.float32)
        blk.train_step_dev(bo.contiguous(), bd.contiguous(), bb.contiguous(), bd.contiguous(), bl.contiguous(),
                           bo.shape[0], lf)
        blk.synchronize()
        losses.append(float(lf[0]))
    assert len(losses) == 3 and all(np.isfinite(losses))
    blk.close()
