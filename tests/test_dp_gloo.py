"""Multi-process (world_size 2, gloo, CPU) coverage of the data-parallel host logic in nerf_amd/dp.py:
the collective is the same torch.distributed call the GPU path makes on RCCL; the per-rank engine here is
the CPU oracle, so what is tested is sharding + all-reduce + replicated Adam, not the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """engine stand-in with the NerfBlock training interface, on the CPU oracle"""

    def __init__(self, c, params):
        from oracle.oracle import Oracle
        self.o = Oracle(c)
        self.p = params.copy()
        self.m = np.zeros_like(params)
        self.v = np.zeros_like(params)
        self.g = torch.zeros(params.size)
        self.t = 0

    def train_backward(self, ro, rd, b, label):
        r = self.o.train_grads(self.p, ro, rd, b, rd, label)
        self.g.copy_(torch.from_numpy(r["grads"]))
        return r["loss_fine"], r["loss_coarse"]

    def optimizer_step(self, scale):
        self.t += 1
        self.o.adam(self.p, self.g.numpy(), self.m, self.v, self.t, scale)

    # the checkpoint interface of NerfBlock that DataParallel.broadcast_state uses
    def get_params(self):
        return self.p.copy()

    def set_params(self, p):
        self.p = np.array(p, np.float32)

    def get_opt_state(self):
        return self.m.copy(), self.v.copy(), self.t

    def set_opt_state(self, m, v, t):
        self.m, self.v, self.t = np.array(m, np.float32), np.array(v, np.float32), int(t)


def _worker(rank, world, port, reduce, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_amd.dp import DataParallel, host_staged_all_reduce, split_batch, row_block
    c = H.CONFIGS["small_dir"]
    n = 32
    ro, rd, b, label = H.make_rays(n, 3)
    eng = OracleEngine(c, H.make_params(c, 1 + rank))        # replicas differ until rank 0's state is broadcast
    staged = reduce == "mean_staged"                        # the exchange through an explicit host copy (test_gpu_dp2)
    dp = DataParallel(eng, lambda: eng.g, reduce="mean" if staged else reduce,
                      all_reduce=host_staged_all_reduce() if staged else None)
    eng.m[:] = rank; eng.t = 5 * rank
    dp.broadcast_state(eng)
    assert eng.t == 0 and not eng.m.any()
    assert dp.mean_scalar(float(rank)) == 0.5
    i0, i1 = split_batch(n, rank, world)
    losses = []
    for _ in range(2):
        losses.append(dp.train_step(lambda: eng.train_backward(ro[i0:i1], rd[i0:i1], b[i0:i1], label[i0:i1])))
    # ray-sharded "render": each rank produces its row block, gathered in row-major order
    Himg, Wimg = 7, 5
    r0, r1 = row_block(Himg, rank, world)
    rows = torch.arange(r0 * Wimg, r1 * Wimg, dtype=torch.float32).reshape(r1 - r0, Wimg, 1).repeat(1, 1, 3)
    img = dp.gather_rows(rows, Himg)
    # bench.py's sharded-frame leg (BASELINE configs[2]) with a stand-in block: uint8 row blocks of unequal height from
    # "the pose", stacked in rank order through the same gather
    from nerf_amd.dp import render_sharded_frame

    class RowsOnly:
        def render_pose_rows(self, Hh, Ww, focal, c2w, near, far, ndc, a, z):
            return (np.arange(a * Ww * 3, z * Ww * 3) % 251).astype(np.uint8).reshape(z - a, Ww, 3)
    frame = render_sharded_frame(RowsOnly(), Himg, Wimg, 1.0, None, 2.0, 6.0, False, rank, world,
                                 lambda r: dp.gather_rows(torch.from_numpy(r), Himg).numpy())
    assert frame.dtype == np.uint8 and frame.shape == (Himg, Wimg, 3)
    np.testing.assert_array_equal(frame.ravel(), (np.arange(Himg * Wimg * 3) % 251).astype(np.uint8))
    out[rank] = (eng.p.copy(), losses, img.numpy())
    dist.destroy_process_group()


def _run(reduce):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, reduce, out), nprocs=world, join=True)
    return out


@pytest.mark.parametrize("reduce", ["mean", "sum", "mean_staged"])
def test_two_rank_training_equals_single_big_batch(reduce):
    out = _run(reduce)
    p0, l0, img0 = out[0]
    p1, l1, img1 = out[1]
    np.testing.assert_array_equal(p0, p1)            # replicas stay identical
    np.testing.assert_array_equal(img0, img1)
    np.testing.assert_array_equal(img0[..., 0].ravel(), np.arange(35, dtype=np.float32))  # row-major pixel order kept
    # reference: one process, whole batch.  mean-of-shard-means == whole-batch mean for equal shards
    c = H.CONFIGS["small_dir"]
    n = 32
    ro, rd, b, label = H.make_rays(n, 3)
    eng = OracleEngine(c, H.make_params(c, 1))
    for _ in range(2):
        eng.train_backward(ro, rd, b, label)
        eng.optimizer_step(2.0 if reduce == "sum" else 1.0)   # "sum" of 2 shard-mean grads = 2 x full-batch grad
    np.testing.assert_allclose(p0, eng.p, rtol=0, atol=2e-6)


def test_row_blocks_cover_image():
    from nerf_amd.dp import row_block
    for Himg in (1, 7, 100, 800):
        for world in (1, 2, 3, 8):
            rows = [row_block(Himg, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == Himg
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1
