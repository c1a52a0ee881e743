"""LLFF (forward-facing) scenes for the hot path: the host side of nerf/llff.scala and of the
"llff" branch of nerf/getDataSet.scala.

Mirrors, by name and meaning (all arithmetic in the dtype of posesBounds.npy -- float64 as LLFF writes it -- and cast
to fp32 at the end, as llff.scala:245 does):
  normalize, cross, inv  <- llff.scala:17-59.  `inv` is the reference's own 3x3 Gauss-Jordan: the first pivot is the
                            row with the largest (signed) first element; the second pivot search compares row 2 with
                            itself (llff.scala:48), i.e. never swaps -- kept as is
  recenter               <- llff.scala:61-94: new world frame from the summed camera z and y axes, positions centred
  get_render_poses       <- llff.scala:133-156: spiral of N poses, `rots` turns, looking at (0, 0, -focal); as in the
                            reference x = cross(up, z) and y = cross(z, x) are not normalised
  load_data              <- llff.scala:163-205: posesBounds.npy [n][17] -> poses [n][3][5] + bds [n][2]; images from
                            images_<factor>/ (sorted by path), or images/ reduced by `factor` when that directory is
                            missing (the reference also writes the reduced PNGs back; this module does not touch disk)
  load_llff_data         <- llff.scala:207-250: axis swap [y, -x, z], scale by 1 / (min(bds) * bdFactor), recenter,
                            spiral render path (rads = 0.9 * max |position|, focal from close/inf depth, dt = .75)
  LlffRays               <- getDataSet.apply, dataSetType == "llff" (getDataSet.scala:28-45,67-171): hwf from pose 0,
                            near = 0.9 min(bds), far = max(bds), every llffHold-th image is test (= val), rays through
                            the device ray generator (NDC: bounds (0, |d'|); else (near |d|, far |d|))

Build definitions (third-party, parity unpinned): image decoding, and DJL's AREA resize restated as the exact box mean
for an integer factor (clamped to [0, 255], llff.scala:190)."""
import math
import os

import numpy as np


def normalize(x):
    return x / np.sqrt((x * x).sum())


def cross(x, y):
    return np.stack([x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]], -1)


def inv(x):
    """llff.scala:37-59 (3x3 Gauss-Jordan on [x | I], see module docstring for the pivot rule)."""
    rows = [r.copy() for r in np.concatenate([x, np.eye(3, dtype=x.dtype)], -1)]
    n0 = int(np.argmax(np.array([rows[0][0], rows[1][0], rows[2][0]])))
    rows[n0], rows[0] = rows[0], rows[n0]
    rows[0] = rows[0] / rows[0][0]
    rows[1] = rows[1] - rows[0] * rows[1][0]
    rows[2] = rows[2] - rows[0] * rows[2][0]
    n1 = int(np.argmax(np.array([rows[2][1], rows[2][1]])))          # always 0: no second swap (llff.scala:48)
    rows[n1 + 1], rows[1] = rows[1], rows[n1 + 1]
    rows[1] = rows[1] / rows[1][1]
    rows[0] = rows[0] - rows[1] * rows[0][1]
    rows[2] = rows[2] - rows[1] * rows[2][1]
    rows[2] = rows[2] / rows[2][2]
    rows[0] = rows[0] - rows[2] * rows[0][2]
    rows[1] = rows[1] - rows[2] * rows[1][2]
    return np.stack([rows[0][3:6], rows[1][3:6], rows[2][3:6]], 0)


def recenter(poses):
    """llff.scala:61-94; poses [n][3][4]."""
    z = normalize(poses[..., 2].sum(0))
    y_tmp = poses[..., 1].sum(0)
    x = normalize(cross(y_tmp, z))
    y = cross(z, x)
    new_world = np.stack([x, y, z], -1)
    out = inv(new_world) @ poses
    out[..., 3] = out[..., 3] - out[..., 3].mean(0, keepdims=True)
    return out


def get_render_poses(rads, focal, zrate, rots, N):
    """llff.scala:133-156 -> [N][3][4]."""
    dt = rads.dtype
    thetas = np.linspace(0, np.float32(2 * np.float32(math.pi) * rots), N + 1, dtype=np.float32)[:-1]
    focal_point = np.array([0, 0, -focal], dt)
    up = np.array([0, 1, 0], dt)
    out = []
    for th in thetas:
        th = float(th)
        c = np.array([math.cos(th), -math.sin(th), -math.sin(th * zrate)], dt) * rads
        z = normalize(c - focal_point)
        x = cross(up, z)
        y = cross(z, x)
        out.append(np.stack([x, y, z, c], -1))
    return np.stack(out, 0)


def _read_image(path):
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), np.uint8)
    return a.astype(np.float32)


def _area(img, factor):
    H, W, C = img.shape
    h, w = H // factor, W // factor
    x = img[:h * factor, :w * factor].reshape(h, factor, w, factor, C)
    return np.clip(x.mean((1, 3), dtype=np.float32), 0, 255)


def load_data(datadir, factor):
    """llff.scala:163-205 -> poses [n][3][5], bds [n][2], images [n][H][W][3] fp32 in [0,1]."""
    arr = np.load(os.path.join(datadir, "posesBounds.npy"), allow_pickle=False)
    poses = arr[:, :-2].reshape(-1, 3, 5).copy()
    bds = arr[:, -2:].copy()
    image_dir = os.path.join(datadir, "images" if factor == 1 else f"images_{factor}")
    if os.path.isdir(image_dir):
        files = sorted(os.path.join(image_dir, f) for f in os.listdir(image_dir))
        images = np.stack([_read_image(f) / np.float32(255) for f in files], 0)
    else:
        src = os.path.join(datadir, "images")
        files = sorted(os.path.join(src, f) for f in os.listdir(src))
        images = np.stack([_area(_read_image(f), factor) / np.float32(255) for f in files], 0)
    poses[:, 0, 4] = images.shape[1]
    poses[:, 1, 4] = images.shape[2]
    poses[:, 2, 4] = poses[:, 2, 4] / factor
    return poses, bds, images


def load_llff_data(datadir, factor=8, bd_factor=.75):
    """llff.scala:207-250 -> (poses [n][3][5], render_poses [120][3][4], images, bds), fp32."""
    poses, bds, images = load_data(datadir, factor)
    poses = np.concatenate([poses[..., 1:2], -poses[..., :1], poses[..., 2:]], -1)
    sc = 1.0 if bd_factor == 0 else 1.0 / (bds.min() * bd_factor)
    poses[:, :3, 3] = poses[:, :3, 3] * sc
    bds = bds * sc
    poses[:, :3, :4] = recenter(poses[:, :3, :4])
    rads = np.abs(poses[:, :3, 3]).max(0) * .9
    close_depth = bds.min() * .9
    inf_depth = bds.max() * 5
    dt = np.float32(.75)
    focal = 1.0 / ((1 - dt) / close_depth + dt / inf_depth)
    render = get_render_poses(rads, focal, np.float32(.5), 2, 120)
    return poses.astype(np.float32), render.astype(np.float32), images, bds.astype(np.float32)


class LlffRays:
    """getDataSet.apply for dataSetType == "llff" (getDataSet.scala:28-45,67-171)."""

    def __init__(self, block, datadir, factor=8, llff_hold=8, ndc=True):
        import torch
        poses5, render_poses, images, bds = load_llff_data(datadir, factor, .75)
        poses = poses5[..., :-1]
        H, W, focal = (float(v) for v in poses5[0, :, -1])
        self.H, self.W, self.focal = int(H), int(W), focal
        self.near = float(np.float32(bds.min()) * np.float32(.9))
        self.far = float(bds.max())
        self.ndc = bool(ndc)
        self.block = block
        n = poses.shape[0]
        i_test = list(range(0, n, llff_hold))
        i_train = [i for i in range(n) if i not in i_test]
        self.i_train, self.i_test = i_train, i_test
        self.splits = {}
        for name, ids in (("train", i_train), ("test", i_test), ("val", i_test)):
            o, d, b = self._rays(poses[ids])
            lab = torch.from_numpy(np.ascontiguousarray(images[ids].reshape(-1, 3))).cuda()
            self.splits[name] = (o, d, b, lab)
        self.render = self._rays(render_poses)
        self.poses, self.render_poses, self.bds = poses, render_poses, bds
        self.split_poses = {"train": poses[i_train], "test": poses[i_test], "val": poses[i_test]}

    def _rays(self, poses):
        import torch
        n = self.H * self.W
        k = len(poses)
        o = torch.empty(k * n, 3, device="cuda"); d = torch.empty(k * n, 3, device="cuda"); b = torch.empty(k * n, 2, device="cuda")
        for i, p in enumerate(poses):
            self.block.k_gen_rays(self.H, self.W, self.focal, np.ascontiguousarray(p, np.float32), self.near, self.far,
                                  self.ndc, o[i * n:(i + 1) * n], d[i * n:(i + 1) * n], b[i * n:(i + 1) * n])
        self.block.synchronize()
        return o, d, b

    def batches(self, split, batch_num, seed=None):
        from .blender import RayBatches
        o, d, b, lab = self.splits[split]
        return RayBatches((o, d, b), lab, batch_num, seed)
