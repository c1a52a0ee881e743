"""BASELINE.json configs that are parity cases rather than bench lines.

configs[0]  "Tiny NeRF, synthetic Blender-format scene 100x100, 64 coarse samples/ray, no hierarchical pass,
            reference CPU path": a whole 100x100 frame from its pose on the GPU against the CPU oracle.
configs[2]  "Full NeRF render of 800x800 Blender-style scene, rays sharded across 8 GPUs": the row-block sharding
            of nerf_amd/dp.py -- eight ranks' blocks rendered one after the other on the one GPU here, stitched in
            rank order -- equals the unsharded frame bit for bit (rays are independent; no collective on this path).
configs[1], [3], [4] are the bench legs (bench.py) and have their parity tests in test_gpu_render / _train / _bf16."""
import math

import numpy as np
import pytest

import gpu_util as G
import helpers as H
from nerf_amd import blender as B
from nerf_amd.dp import render_sharded_frame, row_block
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

ANGLE = 0.6911112070083618


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_config0_tiny_frame_matches_cpu_path(dtype):
    c = H.cfg(D=4, W=64, skips=[2], posL=6, dirL=2, NSamples=64, NImportance=0, useHierarchical=False)
    HW = 100
    focal = 0.5 * HW / math.tan(0.5 * ANGLE)
    pose = B.get_render_poses(0.9, math.pi / 6, 4.0)
    o = Oracle(c)
    params = H.make_params(c, 5)
    ro, rd, bd = o.gen_rays(HW, HW, focal, pose, 2.0, 6.0, False)
    ref = o.forward(params, ro, rd, bd, rd)["rgb_fine"]                       # the CPU path, 10 000 rays
    blk = G.make_block(c, max_rays=4096, dtype=dtype)
    blk.set_params(params)
    try:
        img = blk.render_pose(HW, HW, focal, pose, 2.0, 6.0, False)           # rays generated on the device
        rgb = blk.evaluate(ro, rd, bd, rd)
        np.testing.assert_allclose(rgb, ref, rtol=0, atol=2e-4)
        np.testing.assert_array_equal(img.reshape(-1, 3), o.to_u8(rgb))       # pixel (j, i) at j*W + i, x255 truncated
        want = o.to_u8(ref).astype(np.int16)
        diff = np.abs(img.reshape(-1, 3).astype(np.int16) - want)
        assert diff.max() <= 1 and (diff == 0).mean() > 0.99                  # a 2e-4 difference can cross an integer
    finally:
        blk.close()


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16"])
def test_config2_row_sharded_frame_equals_whole_frame(dtype):
    c = dict(H.CONFIGS["full"], dtype=dtype)
    Himg, Wimg, world = 800, 16, 8                 # all 800 rows (the sharded axis) of a 16-pixel-wide strip
    focal = 0.5 * 800 / math.tan(0.5 * ANGLE)
    pose = B.get_render_poses(2.1, math.pi / 6, 4.0)
    blk = G.make_block(c, max_rays=4096)
    blk.init_params(3)
    try:
        n = Himg * Wimg
        ro, rd, bd = G.empty(n, 3), G.empty(n, 3), G.empty(n, 2)
        blk.k_gen_rays(Himg, Wimg, focal, pose, 2.0, 6.0, False, ro, rd, bd)
        blk.synchronize()
        ro, rd, bd = ro.cpu().numpy(), rd.cpu().numpy(), bd.cpu().numpy()
        whole = blk.render_u8(ro, rd, bd, rd).reshape(Himg, Wimg, 3)
        parts, rows = [], 0
        for rank in range(world):
            r0, r1 = row_block(Himg, rank, world)
            assert r1 - r0 == 100 and r0 == rows                             # contiguous blocks of 100 rows
            s = slice(r0 * Wimg, r1 * Wimg)
            parts.append(blk.render_u8(ro[s], rd[s], bd[s], rd[s]).reshape(r1 - r0, Wimg, 3))
            rows = r1
        np.testing.assert_array_equal(np.concatenate(parts, 0), whole)
    finally:
        blk.close()


@pytest.mark.parametrize("dtype,world", [("f32", 8), ("bf16", 8), ("f16x3", 3)])
def test_config2_pose_rows_of_a_faked_world_stack_to_the_whole_frame(dtype, world):
    """bench.py's `render_sharded_800` leg (nerf_amd.dp.render_sharded_frame over nerfhip_render_pose_rows_u8): every rank
    of a faked world renders its row block of ONE frame straight from the pose; stacked in rank order the blocks equal the
    single-call frame bit for bit -- including a world that does not divide the rows (800 = 267 + 267 + 266) and blocks
    larger than max_rays."""
    c = dict(H.CONFIGS["full"], dtype=dtype)
    Himg, Wimg = 800, 12
    focal = 0.5 * 800 / math.tan(0.5 * ANGLE)
    pose = B.get_render_poses(1.3, math.pi / 6, 4.0)
    blk = G.make_block(c, max_rays=1000)
    blk.init_params(4)
    try:
        whole = blk.render_pose(Himg, Wimg, focal, pose, 2.0, 6.0, False)
        parts = [render_sharded_frame(blk, Himg, Wimg, focal, pose, 2.0, 6.0, False, rank, world) for rank in range(world)]
        assert [p.shape[0] for p in parts] == [row_block(Himg, r, world)[1] - row_block(Himg, r, world)[0] for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(parts, 0), whole)
        assert blk.render_pose_rows(Himg, Wimg, focal, pose, 2.0, 6.0, False, 5, 5).shape == (0, Wimg, 3)   # a rank without rows
        from nerf_amd import NerfHipError
        with pytest.raises(NerfHipError):
            blk.render_pose_rows(Himg, Wimg, focal, pose, 2.0, 6.0, False, 10, Himg + 1)
    finally:
        blk.close()


def test_ndc_pose_rows_stack_to_the_whole_frame_and_equal_explicit_rays():
    """The same sharding for a forward-facing scene (BASELINE configs[4]'s rays: ndc = true, llff.scala:96-132): the row
    blocks of a world that does not divide the rows equal the single-call frame, which equals the frame rendered from the
    rays the device generator writes out (pixel (j, i) of a block is pixel (row0 + j, i) of the frame)."""
    c = dict(H.CONFIGS["full"], NImportance=32)
    Himg, Wimg, world = 37, 20, 3
    focal = 0.9 * Wimg
    pose = np.array([[1, 0, 0, 0.05], [0, 1, 0, 0.02], [0, 0, 1, 0.1]], np.float32)
    blk = G.make_block(c, max_rays=256)
    blk.init_params(6)
    try:
        whole = blk.render_pose(Himg, Wimg, focal, pose, 0.0, 1.0, True)
        parts = [render_sharded_frame(blk, Himg, Wimg, focal, pose, 0.0, 1.0, True, rank, world) for rank in range(world)]
        np.testing.assert_array_equal(np.concatenate(parts, 0), whole)
        n = Himg * Wimg
        ro, rd, bd = G.empty(n, 3), G.empty(n, 3), G.empty(n, 2)
        blk.k_gen_rays(Himg, Wimg, focal, pose, 0.0, 1.0, True, ro, rd, bd)
        blk.synchronize()
        ro, rd, bd = ro.cpu().numpy(), rd.cpu().numpy(), bd.cpu().numpy()
        o_ro, o_rd, o_bd = Oracle(c).gen_rays(Himg, Wimg, focal, pose, 0.0, 1.0, True)
        np.testing.assert_array_equal(ro, o_ro); np.testing.assert_array_equal(rd, o_rd); np.testing.assert_array_equal(bd, o_bd)
        np.testing.assert_array_equal(blk.render_u8(ro, rd, bd, rd).reshape(Himg, Wimg, 3), whole)
        assert whole.std() > 0                                                    # a picture, not a constant
    finally:
        blk.close()
