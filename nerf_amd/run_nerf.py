"""The caller of the hot path: the host loop of nerf/runNerf.scala (train, :22-179; renderToImage, :181-204) over
NerfBlock, so that a scene directory trains, logs and renders the way the reference's `runNerf.main` does.

Mirrors, by name and meaning:
  RunConfig        <- the non-arithmetic fields of nerfConfig (nerfConfig.scala:8-18,45-55): dataSetType, factor,
                      llffHold, halfRes, testSkip, dataDir, logDir, iPrint, iImage, iWeight, iTestSet, iVideo, NIter
  get_data_set     <- getDataSet.apply (getDataSet.scala:23-171); an unknown dataSetType is an error (:69)
  render_to_image  <- runNerf.renderToImage (:181-204): evaluate, x255, truncate to uint8, row-major pixels.  The
                      reference sends one image row per call; here one call per image from its pose (the rays are the
                      same bits, tests/test_gpu_render.py) -- rows are independent, so the image is the same
  train            <- runNerf.train (:87-176): epochs of shuffled batches; every step forward + loss (fine + coarse when
                      hierarchical, :72-73) + backward + Adam; every iPrint steps the mean loss; every iImage steps
                      validation image (idx / iImage - 1) mod n_val as PNG; every iWeight steps the weights; every
                      iTestSet steps the mean fine loss over the test set with the TRAINING-mode forward (the reference
                      calls trainer.forward there, :144, so raw noise and the coarse colour head are on); every iVideo
                      steps all render poses as PNGs; stops after NIter steps
Differences, on purpose: log lines also go to `log` (the reference redirects stdout into logDir/log.txt, :84-85);
weights are saved as .npz with the Adam state (NerfBlock.save_checkpoint) instead of DJL's parameter stream.
Several devices (`config.device` with more than one entry, `next.split(devices)`, :95): one process per GPU, each
calls train(..., dp=DataParallel.for_block(block), seed=<same on every rank>) after dp.shared_stream(block); every
rank draws the same permutation, takes its slice of each batch (split_batch), the gradients are all-reduced once per
step (nerf_amd/dp.py), rank 0 alone writes logs, images and weights; the logged images and video frames are rendered by
ALL ranks, a contiguous block of rows each (render_to_image with dp: BASELINE configs[2], runNerf.scala:187-199).
"""
import os
from dataclasses import dataclass

import numpy as np

from .block import NerfBlock
from .config import NerfConfig


@dataclass
class RunConfig:
    dataSetType: str = "blender"     # nerfConfig.scala:8
    factor: int = 8                  # :10 (llff)
    llffHold: int = 8                # :11
    halfRes: bool = True             # :13 (blender)
    testSkip: int = 1                # :14
    dataDir: str = "./data/nerf_synthetic/lego"   # :45
    logDir: str = "./logs/lego"      # :46
    iPrint: int = 100                # :48
    iImage: int = 500                # :49
    iWeight: int = 50000             # :50
    iTestSet: int = 50000            # :51
    iVideo: int = 100000             # :52
    NIter: int = 500000              # :54


def get_data_set(block, cfg: NerfConfig, run: RunConfig):
    if run.dataSetType == "blender":
        from .blender import BlenderRays
        return BlenderRays(block, run.dataDir, half_res=run.halfRes, test_skip=run.testSkip, white_bkgd=cfg.whiteBkgd)
    if run.dataSetType == "llff":
        from .llff import LlffRays
        return LlffRays(block, run.dataDir, factor=run.factor, llff_hold=run.llffHold, ndc=cfg.ndc)
    raise ValueError(f"Dataset type {run.dataSetType!r} not supported")          # getDataSet.scala:69


def render_to_image(block, ds, poses, dp=None):
    """runNerf.renderToImage for the images of `poses` ([P][3][4]) of data set `ds` -> uint8 [P][H][W][3].
    With a multi-rank `dp` this is a COLLECTIVE: every rank renders its contiguous row block of every frame from the pose
    (runNerf.scala:187-199 evaluates the rows one by one and stacks them; SURVEY 8e: row blocks per GPU, no collective on
    the data path) and the blocks are gathered in rank order -- the frame every rank returns equals the single-GPU one
    bit for bit, in 1/world of the time."""
    if dp is None or dp.world == 1:
        return np.stack([block.render_pose(ds.H, ds.W, ds.focal, p, ds.near, ds.far, ds.ndc) for p in poses], 0)
    from .dp import render_sharded_frame
    dev = f"cuda:{block.config.device}"
    return np.stack([render_sharded_frame(block, ds.H, ds.W, ds.focal, p, ds.near, ds.far, ds.ndc, dp.rank, dp.world,
                                          lambda rows: dp.gather_rows_host(rows, ds.H, dev)) for p in poses], 0)


def _save_png(path, u8):
    from PIL import Image
    Image.fromarray(u8, "RGB").save(path, "PNG")


def eval_test_set(block, ds, batch_num, i_print, log):
    """runNerf.scala:132-157: mean over test batches of the fine-output L2 loss, training-mode forward."""
    total = part = np.float32(0)
    k = 0
    for (o, d, b), lab in ds.batches("test", batch_num):
        fine, _ = block.forward(o.cpu().numpy(), d.cpu().numpy(), b.cpu().numpy(), d.cpu().numpy(), training=True)
        diff = lab.cpu().numpy() - fine
        loss = np.float32(np.mean(diff * diff, dtype=np.float32))               # Loss.l2Loss("L2Loss", 1): mean square
        total += loss
        part += loss
        k += 1
        if k % i_print == 0:
            log(f"{k} iterators test: loss is {part / np.float32(i_print)}.")
            part = np.float32(0)
    mean = float(total / np.float32(max(k, 1)))
    log(f"Test over, mean loss is {mean}.")
    return mean


def train(cfg: NerfConfig, run: RunConfig, block=None, log=None, seed=None, dp=None):
    """runNerf.train.  Returns {"block", "data", "iters", "train_loss": [(idx, mean)], "test_loss": [(idx, mean)]};
    the caller closes the block.  `dp`: a nerf_amd.dp.DataParallel for this rank's block (module docstring)."""
    if run.dataSetType not in ("blender", "llff"):
        raise ValueError(f"Dataset type {run.dataSetType!r} not supported")
    rank, world = (dp.rank, dp.world) if dp is not None else (0, 1)
    if world > 1 and seed is None:
        raise ValueError("data-parallel training needs a seed: every rank must draw the same permutations")
    if cfg.maxRays < cfg.batchNum:
        cfg.maxRays = cfg.batchNum
    if block is None:
        block = NerfBlock(cfg)
        block.init_params(cfg.seed)
    if dp is not None:
        dp.broadcast_state(block)
    # The device-pointer entry points are asynchronous on the block's stream, and the batches they read are made by
    # torch ops (the per-epoch gathers of RayBatches): both must be ordered on ONE stream.  The library cannot run on
    # the legacy default stream, so torch is moved onto an explicit stream shared with the block for the run (the dp
    # caller has normally done this already; doing it again is harmless) and moved back at the end.
    import torch
    from .dp import shared_stream
    prev_stream = torch.cuda.current_stream(cfg.device)
    shared_stream(block, device=cfg.device)
    ds = get_data_set(block, cfg, run)
    dirs = {k: os.path.join(run.logDir, k) for k in ("imageLogs", "weightLogs", "videoLogs")}
    if rank == 0:
        for p in dirs.values():
            os.makedirs(p, exist_ok=True)
    logf = open(os.path.join(run.logDir, "log.txt"), "w") if rank == 0 else None

    def say(line):
        if rank != 0:
            return
        logf.write(line + "\n")
        logf.flush()
        if log is not None:
            log(line)

    def step(o, d, b, lab):
        if dp is None:
            block.train_step_dev(o, d, b, d, lab, int(o.shape[0]), loss)             # viewDir = raysD (:99)
            return
        from .dp import split_batch
        i0, i1 = split_batch(int(o.shape[0]), rank, world)                            # next.split(devices, false) (:95)
        if i1 > i0:
            dp.train_step(lambda: block.train_backward_dev(o[i0:i1], d[i0:i1], b[i0:i1], d[i0:i1], lab[i0:i1], i1 - i0, loss))
        else:                                                                          # fewer rays than ranks: contribute zeros
            loss[:] = 0
            dp.train_step(block.zero_grads)

    def mean_over_ranks(x):
        if world == 1:
            return x
        return np.float32(dp.mean_scalar(x, device=f"cuda:{cfg.device}"))             # lossValue / splits.length (:102)

    hier = cfg.useHierarchical
    val_poses = ds.split_poses["val"]
    out = {"block": block, "data": ds, "iters": 0, "train_loss": [], "test_loss": []}
    loss = np.zeros(2, np.float32)
    say("Train start.")
    idx, stop, loss_sum, epoch = 0, False, np.float32(0), 0
    try:
        while not stop:
            batches = ds.batches("train", cfg.batchNum, None if seed is None else seed + epoch)
            epoch += 1
            for (o, d, b), lab in batches:
                if stop:
                    break
                step(o.contiguous(), d.contiguous(), b.contiguous(), lab.contiguous())
                block.synchronize()                                              # lossValue.getFloat() (:102)
                loss_sum += loss[0] + (loss[1] if hier else np.float32(0))
                idx += 1
                if idx % run.iPrint == 0:
                    mean = float(mean_over_ranks(loss_sum) / np.float32(run.iPrint))
                    say(f"{idx} iterators train: loss is {mean}.")
                    out["train_loss"].append((idx, mean))
                    loss_sum = np.float32(0)
                if idx % run.iImage == 0 and len(val_poses):          # every rank: the frame's rows are shared out
                    who = (idx // run.iImage - 1) % len(val_poses)
                    say(f"{idx} iterators: log image.")
                    img = render_to_image(block, ds, val_poses[who:who + 1], dp)[0]
                    if rank == 0:
                        _save_png(os.path.join(dirs["imageLogs"], f"{idx}.png"), img)
                    say("Log over.")
                if rank == 0 and idx % run.iWeight == 0:
                    say(f"{idx} iterators: save weight.")
                    block.save_checkpoint(os.path.join(dirs["weightLogs"], f"{idx}.npz"))
                    say("Log over.")
                if rank == 0 and idx % run.iTestSet == 0:
                    say(f"{idx} iterators: start to test.")
                    out["test_loss"].append((idx, eval_test_set(block, ds, cfg.batchNum, run.iPrint, say)))
                if idx % run.iVideo == 0:                              # every rank, as the image log
                    say(f"{idx} iterators: log video.")
                    path = os.path.join(dirs["videoLogs"], str(idx))
                    if rank == 0:
                        os.makedirs(path, exist_ok=True)
                    for i, p in enumerate(ds.render_poses):
                        img = render_to_image(block, ds, p[None], dp)[0]
                        if rank == 0:
                            _save_png(os.path.join(path, f"{i}.png"), img)
                    say("Log over.")
                if idx % run.NIter == 0:
                    say(f"{idx} iterators: train over.")
                    stop = True
        say("Train over.")
    finally:
        if logf is not None:
            logf.close()
        block.synchronize()                      # queued work may still read this epoch's tensors
        block.set_stream(None)                   # back to the library's own stream
        torch.cuda.set_stream(prev_stream)
    out["iters"] = idx
    return out


def main(argv=None):
    import argparse
    import json
    ap = argparse.ArgumentParser(description="runNerf.main over the MI355X hot path")
    ap.add_argument("--config", help="JSON with nerfConfig field names (both the arithmetic and the run fields)")
    ap.add_argument("--data-dir")
    ap.add_argument("--log-dir")
    ap.add_argument("--dtype", choices=("f32", "bf16"))
    a = ap.parse_args(argv)
    d = {}
    if a.config:
        with open(a.config) as f:
            d = json.load(f)
    cfg = NerfConfig.from_dict(d)
    run = RunConfig(**{k: v for k, v in d.items() if k in RunConfig.__dataclass_fields__})
    if a.data_dir:
        run.dataDir = a.data_dir
    if a.log_dir:
        run.logDir = a.log_dir
    if a.dtype:
        cfg.dtype = a.dtype
    res = train(cfg, run, log=print)
    res["block"].close()


if __name__ == "__main__":
    main()
