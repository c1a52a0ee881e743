"""Blender-format scenes for the hot path: the host side of nerf/blender.scala and of the
"blender" branch of nerf/getDataSet.scala, so that a scene directory goes straight into NerfBlock.

Mirrors, by name and meaning:
  read_transforms      <- blender.reader (blender.scala:21-40): camera_angle_x + frames[(file_path, rotation, 4x4)]
  get_render_poses     <- blender.getRenderPoses (blender.scala:42-64): theta @ phi @ origin, fp32
  load_blender_data    <- blender.loadBlenderData (blender.scala:110-147): splits train/val/test, testSkip, halfRes,
                          focal = .5 W / tan(.5 camera_angle_x), 40 render poses on radius 4 at phi = pi/6
  BlenderRays          <- getDataSet.apply, dataSetType == "blender" (getDataSet.scala:46-65,67-171): white-background
                          blend images[..., :3] * a + (1 - a), near = 2, far = 6, rays of every pose through the
                          library's device ray generator (nerfhip_k_gen_rays == getRaysNp + normalise + bounds,
                          bit-exact in the GPU tests), flattened per split
  RayBatches           <- nerfDataSet (nerfDataSet.scala:11-74): one random permutation per epoch, consecutive slices of
                          batchNum rays, the last one short

Build definitions (third-party behaviour the reference leaves to DJL / java.awt, not verifiable here -- parity
unpinned for exactly these two points): (1) PNGs are decoded to 8-bit RGBA / RGB / grey; (2) halfRes uses DJL's
NDImageUtils.resize(..., AREA) -- restated as the exact 2x2 box mean in fp32 (what area interpolation is for an integer
factor), cast back to uint8 by truncation as DJL's resize casts back to the input type, before the /255.
Everything else is plain arithmetic restated op by op in fp32.
"""
import json
import math
import os

import numpy as np


def read_transforms(path):
    """blender.scala:21-40 -> (camera_angle_x, [(file_path, rotation, 4x4 float32), ...])"""
    with open(path, "rb") as f:
        meta = json.loads(f.read())
    frames = [(fr["file_path"], np.float32(fr["rotation"]), np.asarray(fr["transform_matrix"], np.float32))
              for fr in meta["frames"]]
    return np.float32(meta["camera_angle_x"]), frames


def get_render_poses(theta, phi, radius):
    """blender.scala:42-64: camera on the sphere of `radius`, looking at the origin; [3][4] fp32."""
    f = np.float32
    sp, cp = f(math.sin(phi)), f(math.cos(phi))          # Math.sin(phi).toFloat: double sin of the float argument
    st, ct = f(math.sin(theta)), f(math.cos(theta))
    origin = np.array([[0, 0, 1, radius], [1, 0, 0, 0], [0, 1, 0, 0]], f)
    phim = np.array([[cp, 0, -sp], [0, 1, 0], [sp, 0, cp]], f)
    thm = np.array([[ct, -st, 0], [st, ct, 0], [0, 0, 1]], f)
    return ((thm @ phim).astype(f) @ origin).astype(f)


def _read_png(path):
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("L", "RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.getbands() or im.mode == "P" else "RGB")
    a = np.asarray(im, np.uint8)
    return a[..., None] if a.ndim == 2 else a


def _half_res_area(img_u8):
    """2x2 box mean (area interpolation at factor 2), fp32, truncated back to uint8 (see module docstring)."""
    H, W, C = img_u8.shape
    h, w = H // 2, W // 2
    x = img_u8[:2 * h, :2 * w].astype(np.float32).reshape(h, 2, w, 2, C)
    return ((x[:, 0, :, 0] + x[:, 0, :, 1] + x[:, 1, :, 0] + x[:, 1, :, 1]) * np.float32(0.25)).astype(np.uint8)


def load_blender_data(data_dir, half_res=False, test_skip=1):
    """blender.scala:110-147 -> (poses [n,3,4], render_poses [40,3,4], images [n,H,W,C] fp32 in [0,1],
    (H, W, focal), i_split = cumulative image counts after train / val / test)."""
    splits = ("train", "val", "test")
    metas = [read_transforms(os.path.join(data_dir, f"transforms_{s}.json")) for s in splits]
    images, poses, i_split = [], [], []
    H = W = 0
    for s, (_, frames) in zip(splits, metas):
        skip = 1 if (s == "train" or test_skip == 0) else test_skip
        for i in range(0, len(frames), skip):
            img = _read_png(os.path.join(data_dir, frames[i][0] + ".png"))
            H, W = img.shape[0], img.shape[1]
            if half_res:
                img = _half_res_area(img)
            images.append(img.astype(np.float32) / np.float32(255))
            poses.append(frames[i][2][:3])
        i_split.append(len(images))
    if half_res:
        H //= 2
        W //= 2
    focal = np.float32(np.float32(0.5) * np.float32(W) / np.float32(math.tan(0.5 * float(metas[0][0]))))
    render = np.stack([get_render_poses(np.float32(2 * np.float32(math.pi) / 40 * i), np.float32(math.pi) / 6, 4.0)
                       for i in range(40)], 0)
    return np.stack(poses, 0).astype(np.float32), render, np.stack(images, 0), (H, W, float(focal)), i_split


class BlenderRays:
    """getDataSet.apply for dataSetType == "blender": per split the flattened rays (device tensors produced by the
    library's ray generator) and labels.  `block` is any NerfBlock (it only supplies the device kernel)."""

    NEAR, FAR = 2.0, 6.0          # getDataSet.scala:62-63

    def __init__(self, block, data_dir, half_res=False, test_skip=1, white_bkgd=True):
        import torch
        poses, render_poses, images, (H, W, focal), i_split = load_blender_data(data_dir, half_res, test_skip)
        if images.shape[-1] == 4:                                  # getDataSet.scala:55-60
            if white_bkgd:
                alpha = images[..., 3:]
                images = images[..., :3] * alpha + (-(alpha - np.float32(1)))
            else:
                images = images[..., :3]
        self.H, self.W, self.focal = H, W, focal
        self.near, self.far, self.ndc = self.NEAR, self.FAR, False
        self.i_split = i_split
        self.block = block
        idx = {"train": range(0, i_split[0]), "val": range(i_split[0], i_split[1]), "test": range(i_split[1], i_split[2])}
        self.splits = {}
        for name, ids in idx.items():
            ids = list(ids)
            o, d, b = self._rays(poses[ids])
            lab = torch.from_numpy(np.ascontiguousarray(images[ids].reshape(-1, 3))).cuda() if ids else torch.empty(0, 3).cuda()
            self.splits[name] = (o, d, b, lab)
        self.render = self._rays(render_poses)
        self.poses, self.render_poses = poses, render_poses
        self.split_poses = {name: poses[list(ids)] for name, ids in idx.items()}

    def _rays(self, poses):
        import torch
        n = self.H * self.W
        k = len(poses)
        o = torch.empty(k * n, 3, device="cuda"); d = torch.empty(k * n, 3, device="cuda"); b = torch.empty(k * n, 2, device="cuda")
        for i, p in enumerate(poses):
            self.block.k_gen_rays(self.H, self.W, self.focal, p, self.NEAR, self.FAR, False, o[i * n:(i + 1) * n],
                                  d[i * n:(i + 1) * n], b[i * n:(i + 1) * n])
        self.block.synchronize()
        return o, d, b

    def batches(self, split, batch_num, seed=None):
        o, d, b, lab = self.splits[split]
        return RayBatches((o, d, b), lab, batch_num, seed)


class RayBatches:
    """nerfDataSet.scala:11-74: every iteration draws one permutation of all rays (the reference swaps element i
    with a uniformly random j for i = 0..size-1) and yields ceil(size / batchNum) consecutive slices."""

    def __init__(self, data, label, batch_num, seed=None):
        self.data, self.label, self.batch_num = data, label, int(batch_num)
        self.size = int(data[0].shape[0])
        self.total = -(-self.size // self.batch_num)
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return self.total

    def permutation(self):
        if self.size > (1 << 20):    # the sequential swap loop below in Python would take minutes: any permutation serves
            return self.rng.permutation(self.size)
        lst = np.arange(self.size)
        js = self.rng.integers(0, self.size, self.size)
        for i in range(self.size):                      # nerfDataSet.scala:34-39 (not Fisher-Yates: kept as is)
            j = js[i]
            lst[i], lst[j] = lst[j], lst[i]
        return lst

    def __iter__(self):
        import torch
        perm = torch.from_numpy(self.permutation()).to(self.data[0].device)
        data = [x[perm] for x in self.data]
        label = self.label[perm]
        for k in range(self.total):
            s = slice(k * self.batch_num, min((k + 1) * self.batch_num, self.size))
            yield tuple(x[s] for x in data), label[s]
