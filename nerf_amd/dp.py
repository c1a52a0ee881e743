"""Data-parallel training and ray-sharded rendering across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Training couples ranks in
exactly ONE place: the flat fp32 gradient buffer (parameter order, 4.24 MB for the full coarse+fine model)
is all-reduced once per step, between `train_backward` and `optimizer_step`; replicas start from identical
parameters and apply the identical Adam update, so they stay identical.  This replaces the reference's
single-process loop `for (s <- splits) { forward; backward }; trainer.step()` over `next.split(devices)`
(runNerf.scala:95-106), where DJL's parameter server sums the per-device gradients on the first device.

Reduction semantics: the reference gives every device the MEAN loss of its own shard and (DJL 0.16
LocalParameterServer, third-party, recalled) SUMS the device gradients.  Default here is "mean" -- g GPUs x B
rays then equals one device running a g*B batch -- and "sum" reproduces the recalled DJL rule.

Rendering needs no collective: rays are independent, every rank renders a contiguous block of image rows
(`row_block`) so the reference's row-major pixel mapping (runNerf.scala:187-199) is preserved.
"""
import torch
import torch.distributed as dist


class _CudaView:
    """zero-copy torch view of a raw device buffer (CUDA array interface)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def grad_tensor(block):
    """torch view of block's flat gradient buffer (device memory owned by libnerfhip)."""
    ptr, n = block.grad_buffer()
    return torch.as_tensor(_CudaView(ptr, n), device=f"cuda:{block.config.device}")


def shared_stream(*blocks, device=None):
    """Create a torch stream, make it current and hand it to every block.

    The device-pointer entry points (`train_backward_dev`, `optimizer_step`, `render_dev`) are asynchronous, so the
    gradient all-reduce must be ordered against them ON A STREAM: torch's collectives are ordered after the
    current torch stream and make it wait for their result.  The library cannot run on the legacy default stream
    (`set_stream(0)` selects its own non-blocking stream), hence an explicit one.  Returns the torch stream."""
    s = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(s)
    for b in blocks:
        b.set_stream(s.cuda_stream)
    return s


def row_block(H, rank, world):
    """rows [r0, r1) of an H-row image rendered by `rank`: contiguous, sizes differ by at most one row."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def render_sharded_frame(block, H, W, focal, c2w, near, far, ndc=False, rank=0, world=1, gather=None):
    """ONE H x W frame rendered by `world` ranks (BASELINE configs[2]): this rank renders `row_block(H, rank, world)`
    from the pose (rays made on the device, nerfhip_render_pose_rows_u8) -- no collective on the data path -- and
    `gather(rows_u8 [rows][W][3]) -> [H][W][3]` stacks the blocks in rank order (DataParallel.gather_rows; None = return
    this rank's block).  The stacked image equals the single-GPU frame bit for bit (runNerf.scala:187-199: rows are
    evaluated independently and stacked in order)."""
    r0, r1 = row_block(H, rank, world)
    rows = block.render_pose_rows(H, W, focal, c2w, near, far, ndc, r0, r1)
    return rows if gather is None else gather(rows)


def split_batch(n, rank, world):
    """Batch.split(devices, false) (runNerf.scala:95): slice [i0, i1) of an n-ray batch owned by `rank`."""
    return row_block(n, rank, world)


def host_staged_all_reduce(group=None):
    """all-reduce(sum) of a device tensor THROUGH THE HOST: copy down, reduce the host copy, copy back, all ordered on
    torch's current stream.  For groups whose backend only carries host tensors reliably (gloo); RCCL needs no such
    staging."""
    def fn(t):
        h = t.cpu()                      # synchronises with the current stream: the backward kernels have finished
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)                       # H2D on the current stream, ahead of the optimizer step queued behind it
    return fn


class DataParallel:
    """Wraps an engine with train_backward(...)->losses, grad_view()->1-D tensor, optimizer_step(scale).

    `engine` is a nerf_amd.NerfBlock on GPU boxes (see `for_block`); the CPU tests drive the same class with a
    stand-in engine over gloo, so the collective logic is covered without a GPU."""

    def __init__(self, engine, grad_view, group=None, reduce="mean", all_reduce=None):
        """all_reduce: optional callable(tensor) that sums the tensor over the ranks in place, used instead of
        dist.all_reduce -- e.g. `host_staged_all_reduce` when the group's backend cannot carry device tensors."""
        if reduce not in ("mean", "sum"):
            raise ValueError("reduce must be 'mean' or 'sum'")
        self.engine = engine
        self.grad_view = grad_view
        self.group = group
        self.reduce = reduce
        self.all_reduce = all_reduce
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @staticmethod
    def for_block(block, group=None, reduce="mean", all_reduce=None):
        g = grad_tensor(block)
        return DataParallel(block, lambda: g, group, reduce, all_reduce)

    def train_step(self, backward_fn):
        """backward_fn() runs forward+loss+backward on this rank's shard and returns its losses."""
        losses = backward_fn()
        if self.world > 1:
            if self.all_reduce is not None:
                self.all_reduce(self.grad_view())
            else:
                dist.all_reduce(self.grad_view(), op=dist.ReduceOp.SUM, group=self.group)
        self.engine.optimizer_step(1.0 / self.world if self.reduce == "mean" else 1.0)
        return losses

    def mean_scalar(self, x, device=None):
        """mean of a host scalar over the ranks (the loss the run loop prints, runNerf.scala:102)"""
        if self.world == 1:
            return x
        on_dev = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([float(x)], dtype=torch.float64, device=device if on_dev else "cpu")
        dist.all_reduce(t, group=self.group)
        return float(t.item()) / self.world

    def _bcast(self, a, src):
        """broadcast a numpy array through whatever the group's backend can carry: RCCL ("nccl") only moves device
        tensors, gloo moves host tensors"""
        if self.world == 1:
            return a
        t = torch.from_numpy(a)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def broadcast_params(self, get_params, set_params, src=0):
        """make replicas identical (the reference initialises once and copies to every device)"""
        set_params(self._bcast(get_params(), src))

    def broadcast_state(self, block, src=0):
        """parameters AND Adam moments / update count of rank `src` to every rank, so that a resume where only one
        rank loaded the checkpoint cannot diverge"""
        import numpy as np
        block.set_params(self._bcast(block.get_params(), src))
        m, v, step = block.get_opt_state()
        st = self._bcast(np.array([step], np.int64), src)
        block.set_opt_state(self._bcast(m, src), self._bcast(v, src), int(st[0]))

    def gather_rows_host(self, rows_u8, H, device=None):
        """`gather_rows` for a numpy row block (what nerfhip_render_pose_rows_u8 hands back): staged through a device tensor
        when the group's backend only carries those (RCCL), through a host tensor otherwise (gloo); numpy [H][W][3] out"""
        if self.world == 1:
            return rows_u8
        t = torch.from_numpy(rows_u8)
        if dist.get_backend(self.group) == "nccl":
            t = t.to(device if device is not None else "cuda")
        return self.gather_rows(t, H).cpu().numpy()

    def gather_rows(self, local_rows, H):
        """all-gather row blocks of a rendered image (uint8 or float tensor [rows, W, 3]) on every rank"""
        if self.world == 1:
            return local_rows
        sizes = [row_block(H, r, self.world)[1] - row_block(H, r, self.world)[0] for r in range(self.world)]
        mx = max(sizes)   # all_gather wants equal shapes: pad the shorter blocks, trim after
        mine = local_rows.contiguous()
        if mine.shape[0] < mx:
            mine = torch.cat([mine, mine.new_zeros((mx - mine.shape[0],) + tuple(mine.shape[1:]))], 0)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        parts = [p[:k] for p, k in zip(parts, sizes)]
        return torch.cat(parts, 0)
