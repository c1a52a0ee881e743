#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native NeRF hot path (libnerfhip.so).

Workload (BASELINE.json configs[1] / configs[3]): full NeRF coarse+fine (64+128 samples/ray), 8x256 MLP with
skip at 4, posL 10 / dirL 4 view-direction head, synthetic Blender-style 400x400 scene, fp32 MFMA.
  * a "step" = ONE training iteration (forward + L2 loss + backward + [grad all-reduce] + Adam) on
    4096 rays per GPU (weak scaling; runNerf.scala:94-106), inputs already resident in HBM;
  * after the timed train steps the same process renders 400x400 frames (160 000 rays each) and reports
    render rays/s in the "render" object of the same JSON line.
`value` is whole-job train-step rays/s (all ranks).  Launch: python bench.py --gpus N.  For N > 1 one process per GPU
is needed (RCCL all-reduce of the flat gradient buffer): under torch.distributed.run (WORLD_SIZE set) this process IS
one of the ranks; started plainly, it launches the N ranks itself (torch.distributed.run as a child process, before
anything here touches the GPU), passes their single JSON line through and exits with their code.  A world size that
differs from --gpus is an error, never a line with another n_gpus.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16x3": 833.3}   # MI355X_MICROARCH.md: dense MFMA peaks (f16x3: the f16 peak / 3 MFMAs per product)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (runNerf.scala:95 `next.split(trainer.getDevices)`: the job owns N
    devices): start N ranks of this same script under torch.distributed.run, one per GPU.  The parent never imports torch
    and never touches the GPU; the children inherit stdout, so rank 0's JSON line is this process's JSON line."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args, world, rank):
    """--dry-launch: the ranks only rendezvous (gloo, no GPU) and rank 0 prints the world it saw -- the launch logic of
    this script under test on a CPU box"""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo") if world > 1 else None
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
    if int(seen.item()) != args.gpus:
        raise SystemExit(f"dry launch: {int(seen.item())} ranks met, --gpus {args.gpus}")
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": int(seen.item()), "world_size_env": world}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="training rays per GPU per step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16x3"])
    ap.add_argument("--render-frames", type=int, default=3)
    ap.add_argument("--hw", type=int, default=400, help="synthetic image side")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-render", action="store_true")
    ap.add_argument("--no-bf16-train", action="store_true", help="skip the bf16 training legs (same batch; config 5)")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="time budget of the whole CPU baseline")
    ap.add_argument("--native-rccl", action="store_true", help="data-parallel exchange through the library's own RCCL "
                    "entry points (nerfhip_dp_*: no torch in the step) instead of torch.distributed.all_reduce on the "
                    "gradient buffer; torch.distributed still provides rendezvous, barriers and the max-over-ranks time")
    ap.add_argument("--extras-at-any-n", action="store_true", help="run the extra single-GPU legs (other dtypes, config 5, "
                    "800x800 frames, HBM kernels) on every rank at N > 1 as well; by default they run at N = 1 only")
    ap.add_argument("--force-dp", action="store_true", help="take the data-parallel code path (backward, all-reduce, "
                    "optimizer step) even with one rank: rehearsal of the N > 1 loop on a 1-GPU box")
    ap.add_argument("--dry-launch", action="store_true", help="launch test without a GPU: the ranks rendezvous over gloo, "
                    "rank 0 prints {\"dry_launch\": true, \"n_gpus\": <ranks that met>} and everything exits")
    ap.add_argument("--no-native-leg", action="store_true", help="skip the short native-RCCL comparison leg that runs "
                    "behind the headline line whenever the step is data parallel")
    ap.add_argument("--native-leg-timeout", type=float, default=180.0, help="seconds before the native-RCCL leg is given up "
                    "(the headline line is already out by then)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N ranks, or an error: never a 1-rank run under an N-GPU request
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))          # nothing above touched the GPU (numpy only)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as "
                         f"{args.gpus} GPUs (start it as `python bench.py --gpus {args.gpus}` or under torch.distributed.run "
                         f"with --nproc-per-node {args.gpus})")
    if args.dry_launch:
        return dry_launch(args, world, rank)

    import torch
    import torch.distributed as dist
    from nerf_amd import NerfBlock, NerfConfig
    from nerf_amd.dp import DataParallel, grad_tensor, render_sharded_frame, row_block

    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local)
    dp = world > 1 or args.force_dp
    if dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries exactly one JSON line,
        # so file descriptor 1 points at stderr until the communicator exists (eager init + one tiny all-reduce)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    def note(msg):   # progress on stderr (stdout carries exactly one JSON line)
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    # one explicit stream for the library's kernels, torch's tensor ops and (through torch) the RCCL collectives:
    # the all-reduce is ordered after the backward kernels and the Adam kernel after the all-reduce on it
    tstream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    HW = args.hw
    n_frame = HW * HW
    B = args.rays
    base = dict(useDir=True, useSH=False, useHierarchical=True, posL=10, dirL=4, D=8, W=256, skips=[4], NSamples=64,
                NImportance=128, rawNoiseStd=0.0, whiteBkgd=True, linDisp=False, perturb=False, lrate=5e-4,
                lrateDecay=250, device=local, dtype=args.dtype, seed=1234)
    trainer = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=B)))
    renderer = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=n_frame)))
    trainer.set_stream(stream)
    renderer.set_stream(stream)
    trainer.init_params(7)                       # same seed on every rank: replicas start identical
    renderer.set_params(trainer.get_params())

    # ---- synthetic scene: camera on a sphere of radius 4 looking at the origin (blender.scala:142-145)
    focal = 0.5 * HW / math.tan(0.5 * 0.6911112070083618)
    th = 0.7 + 0.5 * rank
    phi = math.pi / 6
    cam = 4.0 * np.array([math.cos(th) * math.cos(phi), math.sin(th) * math.cos(phi), math.sin(phi)])
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.stack([right, up, -fwd, cam], 1).astype(np.float32)      # [3][4]
    f32 = dict(dtype=torch.float32, device="cuda")
    ro, rd, bd = torch.empty(n_frame, 3, **f32), torch.empty(n_frame, 3, **f32), torch.empty(n_frame, 2, **f32)
    renderer.k_gen_rays(HW, HW, focal, c2w, 2.0, 6.0, False, ro, rd, bd)
    torch.cuda.synchronize()
    label_full = 0.5 + 0.5 * torch.sin(4.0 * rd + ro)                   # smooth analytic target image
    gsel = torch.Generator(device="cpu").manual_seed(100 + rank)
    idx = torch.randperm(n_frame, generator=gsel)[:B].cuda()
    t_o, t_d, t_b, t_l = ro[idx].contiguous(), rd[idx].contiguous(), bd[idx].contiguous(), label_full[idx].contiguous()
    grad = grad_tensor(trainer) if dp else None   # zero-copy torch view of the flat gradient buffer
    native = dp and args.native_rccl

    def native_init(blk):
        """rank 0's rendezvous id to every rank (carried by the torch group), then ncclCommInitRank inside the library"""
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(NerfBlock.dp_unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, src=0)
        torch.cuda.synchronize()
        sys.stdout.flush(); saved_ = os.dup(1); os.dup2(2, 1)      # RCCL's banner must not land on stdout
        try:
            blk.dp_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
            blk.dp_broadcast_state(0)
        finally:
            sys.stdout.flush(); os.dup2(saved_, 1); os.close(saved_)

    if native:
        native_init(trainer)

    def train_step():
        if not dp:
            trainer.train_step_dev(t_o, t_d, t_b, t_d, t_l, B)
        elif native:
            trainer.dp_train_step_dev(t_o, t_d, t_b, t_d, t_l, B)    # backward + ncclAllReduce + Adam(1/world), one C call
        else:
            trainer.train_backward_dev(t_o, t_d, t_b, t_d, t_l, B)
            dist.all_reduce(grad)                                  # RCCL: one flat 4.24 MB message
            trainer.optimizer_step(1.0 / world)                    # mean over devices

    def barrier():
        if dp:
            dist.barrier()

    note("set up")
    for _ in range(args.warmup):
        train_step()
    torch.cuda.synchronize(); barrier()
    note("warm-up done")
    trainer.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    prof = trainer.get_profile()
    trainer.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    train_rays_s = world * B * args.steps / dt
    note("train leg done")

    # ---- render leg: whole 400x400 frames, ray-parallel, no collective
    out = torch.empty(n_frame, 3, **f32)
    renderer.set_params(trainer.get_params())
    renderer.render_dev(ro, rd, bd, rd, n_frame, out)
    torch.cuda.synchronize(); barrier()
    renderer.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.render_frames):
        renderer.render_dev(ro, rd, bd, rd, n_frame, out)
    torch.cuda.synchronize(); barrier()
    dtr = time.perf_counter() - t0
    rprof = renderer.get_profile()
    if world > 1:
        t = torch.tensor([dtr], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtr = float(t.item())
    render_rays_s = world * n_frame * args.render_frames / dtr
    note("render leg done")

    # ---- BASELINE configs[2]: ONE 800x800 frame, its rows sharded over the ranks (runNerf.scala:187-199: rows are
    # independent evaluate calls stacked in order).  Every rank renders row_block(800, rank, world) of the SAME pose
    # straight from the pose (rays made on the device, uint8 rows to the host: nerfhip_render_pose_rows_u8 -- no collective
    # on the data path), rank order stacks the blocks (all-gather of the uint8 rows, inside the timed region).  At N = 1
    # the block is the whole frame, so the values at N = 1 and N = 8 are the same workload: strong scaling of one frame.
    hw8 = 800
    f8 = 0.5 * hw8 / math.tan(0.5 * 0.6911112070083618)
    cam8 = 4.0 * np.array([math.cos(0.7) * math.cos(phi), math.sin(0.7) * math.cos(phi), math.sin(phi)])
    fwd8 = -cam8 / np.linalg.norm(cam8)
    right8 = np.cross(fwd8, [0, 0, 1.0]); right8 /= np.linalg.norm(right8)
    c2w8 = np.stack([right8, np.cross(right8, fwd8), -fwd8, cam8], 1).astype(np.float32)     # the same pose on every rank
    dpg = DataParallel(renderer, lambda: None) if dp else None

    def gather8(rows):
        if world == 1:
            return rows
        return dpg.gather_rows(torch.from_numpy(rows).cuda(), hw8).cpu().numpy()

    def sharded_frame():
        return render_sharded_frame(renderer, hw8, hw8, f8, c2w8, 2.0, 6.0, False, rank, world, gather8)
    img8 = sharded_frame()                                 # warm-up (first call allocates the block's ray buffers)
    torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    img8 = sharded_frame()
    torch.cuda.synchronize(); barrier()
    dt8 = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt8], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt8 = float(t.item())
    assert img8.shape == (hw8, hw8, 3), img8.shape
    r0_, r1_ = row_block(hw8, rank, world)
    render_sharded = {"metric": "rays/sec (render, one 800x800 frame, rows sharded over the ranks)", "value": round(hw8 * hw8 / dt8, 1),
                      "unit": "rays/s", "workload": "BASELINE configs[2]: ONE 800x800 frame from its pose, contiguous row blocks, "
                      "uint8 rows gathered in rank order", "scaling": "strong", "ms_per_frame": round(1e3 * dt8, 2), "frames": 1,
                      "rows_per_rank": r1_ - r0_, "n_gpus": world, "dtype": args.dtype,
                      "image_checksum": int(img8.astype(np.uint64).sum())}
    note("sharded 800x800 frame done")

    # ---- roofline of the dominant kernel of the timed train region (HIP events on the launch stream)
    kern = {"mlp_fwd": (prof["mlp_fwd_ms"], prof["mlp_fwd_flop"]), "mlp_bwd": (prof["mlp_bwd_ms"], prof["mlp_bwd_flop"]),
            "mlp_wgrad": (prof["mlp_wgrad_ms"], prof["mlp_wgrad_flop"])}
    dom = max(kern, key=lambda k: kern[k][0])
    peak = PEAK_TFLOPS[args.dtype]
    ach = kern[dom][1] / max(kern[dom][0], 1e-9) / 1e9
    roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": None,
                "ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in kern.items()},
                "tflops": {k: round(v[1] / max(v[0], 1e-9) / 1e9, 2) for k, v in kern.items()},
                "other_ms_per_step": round((prof["composite_ms"] + prof["sampling_ms"] + prof["other_ms"]) / args.steps, 4)}
    roofline.update(pmc_traffic(dom, args.dtype))
    if args.dtype == "f16x3" and dom == "mlp_wgrad":
        # the split-f16 weight gradient streams its fp32 operands once per product at the matrix pipe's f16 rate: HBM-bound.
        # Algorithmic bytes per 32-sample column of both networks: (A rows + B rows) x 128 B over the products
        # 256x63, 7 x 256x256, 256x63 (skip), 129x256 (head + density row), 3x128
        rows = (256 + 63) + 7 * (256 + 256) + (256 + 63) + (129 + 256) + (3 + 128)
        alg = rows * 128.0 * (B * (64 + 192) / 32.0)
        gbs = alg / max(kern[dom][0] / args.steps, 1e-9) / 1e6
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                         "algorithmic_bytes": round(alg)})
    r_ach = rprof["mlp_fwd_flop"] / max(rprof["mlp_fwd_ms"], 1e-9) / 1e9
    render = {"metric": "rays/sec (render)", "value": round(render_rays_s, 1), "unit": "rays/s",
              "workload": f"{HW}x{HW} frame, {n_frame} rays/GPU/frame", "frames": args.render_frames,
              "ms_per_frame": round(1e3 * dtr / args.render_frames, 3),
              "samples_per_s": round(render_rays_s * 256, 1),      # network evaluations: 64 coarse + 192 fine
              "roofline": {"kernel": "mlp_fwd", "bound": "mfma", "achieved": round(r_ach, 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(r_ach / peak, 4)}}

    # ---- everything below is extra reporting next to the headline numbers above; with one rank a failure in it is
    # recorded in the line instead of losing the line (with several ranks it must propagate: the others would hang)
    render_bf16 = render_f16x3 = train_f16x3 = train_bf16 = config5 = render_pose = hbm_kernels = cpu = None
    extras_error = None
    extras_skipped = None

    class _ExtrasAtN1Only(Exception):
        pass

    try:
        # the extra legs characterise ONE GPU and do not change with N: at N > 1 they are skipped (every rank would have to
        # run them in step, and a failure in one of them would cost the scaling run its headline line)
        if world > 1 and not args.extras_at_any_n:
            raise _ExtrasAtN1Only()
        # ---- render legs on the other MFMA paths, same frames: bf16 (BASELINE configs[4] arithmetic; parity bar 1e-2 abs,
        # tests/test_gpu_bf16.py) and split-f16 (dtype "f16x3": three f16 MFMAs per product, held to the fp32 parity
        # tests at the fp32 tolerances, tests/test_gpu_render.py) -- reported under their own dtype labels
        def render_leg(dt, label, peak_tf, peak_note):
            rb = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=n_frame, dtype=dt)))
            rb.set_stream(stream)
            rb.set_params(trainer.get_params())
            rb.render_dev(ro, rd, bd, rd, n_frame, out)
            torch.cuda.synchronize(); barrier()
            rb.set_profiling(True)
            t0_ = time.perf_counter()
            for _ in range(args.render_frames):
                rb.render_dev(ro, rd, bd, rd, n_frame, out)
            torch.cuda.synchronize(); barrier()
            dtb = time.perf_counter() - t0_
            bprof = rb.get_profile()
            if world > 1:
                t_ = torch.tensor([dtb], device="cuda", dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                dtb = float(t_.item())
            b_ach = bprof["mlp_fwd_flop"] / max(bprof["mlp_fwd_ms"], 1e-9) / 1e9
            rb.close()
            note(f"{dt} render leg done")
            return {"metric": f"rays/sec (render, {label})", "value": round(world * n_frame * args.render_frames / dtb, 1),
                    "unit": "rays/s", "dtype": dt, "ms_per_frame": round(1e3 * dtb / args.render_frames, 3),
                    "roofline": {"kernel": f"mlp_fwd_{dt}", "bound": "mfma", "achieved": round(b_ach, 2), "peak": peak_tf,
                                 "unit": "TFLOP/s", "frac": round(b_ach / peak_tf, 4), "peak_note": peak_note},
                    "hbm_kernels_ms_per_frame": {"composite": round(bprof["composite_ms"] / args.render_frames, 4),
                                                 "sampling": round(bprof["sampling_ms"] / args.render_frames, 4)}}

        render_bf16 = render_f16x3 = None
        if args.dtype == "f32" and not args.no_bf16_render:
            render_bf16 = render_leg("bf16", "bf16 MFMA", PEAK_TFLOPS["bf16"], "dense bf16 MFMA peak")
            render_f16x3 = render_leg("f16x3", "split-f16 MFMA, fp32-grade", round(PEAK_TFLOPS["bf16"] / 3.0, 1),
                                      "dense f16 MFMA peak / 3 (three MFMAs per fp32-grade product); achieved counts the "
                                      "ALGORITHMIC fp32 FLOPs")

        # ---- bf16-MFMA training legs: the same batch (configs[3] arithmetic in bf16), and BASELINE configs[4]:
        # 1024x1024 forward-facing scene, NDC rays (llff.scala:96-132), 64 + 256 samples per ray
        def timed_train(blk, o_, d_, b_, l_, nsteps, dt="bf16"):
            g = grad_tensor(blk) if dp else None
            if native:
                native_init(blk)

            def one():
                if not dp:
                    blk.train_step_dev(o_, d_, b_, d_, l_, B)
                elif native:
                    blk.dp_train_step_dev(o_, d_, b_, d_, l_, B)
                else:
                    blk.train_backward_dev(o_, d_, b_, d_, l_, B)
                    dist.all_reduce(g)
                    blk.optimizer_step(1.0 / world)
            for _ in range(args.warmup):
                one()
            torch.cuda.synchronize(); barrier()
            blk.set_profiling(True)
            t0_ = time.perf_counter()
            for _ in range(nsteps):
                one()
            torch.cuda.synchronize(); barrier()
            d_t = time.perf_counter() - t0_
            pr = blk.get_profile()
            blk.set_profiling(False)
            if world > 1:
                t_ = torch.tensor([d_t], device="cuda", dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                d_t = float(t_.item())
            ks = {"mlp_fwd": (pr["mlp_fwd_ms"], pr["mlp_fwd_flop"]), "mlp_bwd": (pr["mlp_bwd_ms"], pr["mlp_bwd_flop"]),
                  "mlp_wgrad": (pr["mlp_wgrad_ms"], pr["mlp_wgrad_flop"])}
            return {"value": round(world * B * nsteps / d_t, 1), "unit": "rays/s", "dtype": dt, "steps": nsteps,
                    "ms_per_step": round(1e3 * d_t / nsteps, 3),
                    "kernel_ms_per_step": {k: round(v[0] / nsteps, 4) for k, v in ks.items()},
                    "kernel_tflops": {k: round(v[1] / max(v[0], 1e-9) / 1e9, 1) for k, v in ks.items()}}

        # ---- dtype f16x3 training: split-f16 forward, backward-data and weight-gradient kernels (fp32-grade: held to the
        # fp32 gradient tests at the fp32 tolerances, tests/test_gpu_train.py); own dtype label
        train_f16x3 = None
        if args.dtype == "f32" and not args.no_bf16_train:
            tx = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=B, dtype="f16x3")))
            tx.set_stream(stream)
            tx.init_params(7)
            train_f16x3 = timed_train(tx, t_o, t_d, t_b, t_l, args.steps, "f16x3")
            train_f16x3["metric"] = "rays/sec (train step, split-f16 MFMA forward/backward/weight gradient, fp32-grade)"
            tx.close()
            note("f16x3 train leg done")

        train_bf16 = config5 = None
        if args.dtype == "f32" and not args.no_bf16_train:
            tb = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=B, dtype="bf16")))
            tb.set_stream(stream)
            tb.init_params(7)
            train_bf16 = timed_train(tb, t_o, t_d, t_b, t_l, args.steps)
            train_bf16["metric"] = "rays/sec (train step, bf16 MFMA)"
            tb.close()
            note("bf16 train leg done")
            # config 5
            HW5 = 1024
            n5 = HW5 * HW5
            c5 = dict(base, NImportance=256, dtype="bf16")
            t5 = NerfBlock(NerfConfig.from_dict(dict(c5, maxRays=B)))
            r5 = NerfBlock(NerfConfig.from_dict(dict(c5, maxRays=n5 // 4)))
            t5.set_stream(stream); r5.set_stream(stream)
            t5.init_params(7)
            ro5, rd5, bd5 = torch.empty(n5, 3, **f32), torch.empty(n5, 3, **f32), torch.empty(n5, 2, **f32)
            c2w5 = np.array([[1, 0, 0, 0.05 * rank], [0, 1, 0, 0.02], [0, 0, 1, 0.1]], np.float32)
            r5.k_gen_rays(HW5, HW5, 0.9 * HW5, c2w5, 0.0, 1.0, True, ro5, rd5, bd5)
            torch.cuda.synchronize()
            lab5 = 0.5 + 0.5 * torch.sin(3.0 * rd5 + ro5)
            idx5 = torch.randperm(n5, generator=gsel)[:B].cuda()
            config5 = timed_train(t5, ro5[idx5].contiguous(), rd5[idx5].contiguous(), bd5[idx5].contiguous(),
                                  lab5[idx5].contiguous(), args.steps)
            r5.set_params(t5.get_params())
            out5 = torch.empty(n5, 3, **f32)
            q = n5 // 4

            def frame5():
                for k in range(4):   # a 1024x1024 frame in four 256-row blocks (workspace sized for one block)
                    sl = slice(k * q, (k + 1) * q)
                    r5.render_dev(ro5[sl], rd5[sl], bd5[sl], rd5[sl], q, out5[sl])
            frame5()
            torch.cuda.synchronize(); barrier()
            t0 = time.perf_counter()
            frame5()
            torch.cuda.synchronize(); barrier()
            dt5 = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt5], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt5 = float(t.item())
            config5.update({"metric": "rays/sec (train step, bf16 MFMA, config 5)",
                            "workload": "1024x1024 forward-facing scene, NDC rays, 64+256 samples/ray, 8x256 MLP",
                            "samples_per_ray": 320,
                            "render": {"value": round(world * n5 / dt5, 1), "unit": "rays/s", "ms_per_frame": round(1e3 * dt5, 2),
                                       "frames": 1}})
            t5.close(); r5.close()

        # ---- BASELINE configs[2] per GPU: one 800x800 frame straight from its camera pose (nerfhip_render_pose_u8: rays
        # made on the device, rendered in max_rays blocks, uint8 image copied to the host) -- the PCIe-inclusive rate
        def pose_frame(blk, frames=2):
            hw8 = 800
            f8 = 0.5 * hw8 / math.tan(0.5 * 0.6911112070083618)
            blk.render_pose(hw8, hw8, f8, c2w, 2.0, 6.0, False)
            torch.cuda.synchronize(); barrier()
            t0_ = time.perf_counter()
            for _ in range(frames):
                blk.render_pose(hw8, hw8, f8, c2w, 2.0, 6.0, False)
            d_ = time.perf_counter() - t0_
            if world > 1:
                t_ = torch.tensor([d_], device="cuda", dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                d_ = float(t_.item())
            return {"value": round(world * hw8 * hw8 * frames / d_, 1), "unit": "rays/s", "ms_per_frame": round(1e3 * d_ / frames, 2)}

        render_pose = {"workload": "800x800 frame from (pose,H,W,f), uint8 image to host, per GPU", "f32": pose_frame(renderer, 1)}
        if args.dtype == "f32" and not args.no_bf16_render:
            rb2 = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=n_frame, dtype="bf16")))
            rb2.set_stream(stream)
            rb2.set_params(trainer.get_params())
            render_pose["bf16"] = pose_frame(rb2, 3)
            rb2.close()
            rb3 = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=n_frame, dtype="f16x3")))
            rb3.set_stream(stream)
            rb3.set_params(trainer.get_params())
            render_pose["f16x3"] = pose_frame(rb3, 2)
            rb3.close()

        # ---- HBM-bound kernels on one frame's fine pass (160 000 rays x 192 samples): achieved GB/s on algorithmic bytes
        # (compositing fwd 24 B/sample, bwd 40 B/sample; resampling 8 B per coarse + 4 B per fine sample)
        def timed_kernel(fn, reps=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        S = 192
        gk = torch.Generator(device="cuda").manual_seed(5)
        sig = torch.rand(n_frame, S, generator=gk, **f32) * 3.0
        crw = torch.randn(n_frame, S, 3, generator=gk, **f32)
        zz = torch.sort(torch.rand(n_frame, S, generator=gk, **f32) * 4.0 + 2.0, dim=1).values.contiguous()
        w_, rgb_, dep_, acc_ = torch.empty(n_frame, S, **f32), torch.empty(n_frame, 3, **f32), torch.empty(n_frame, **f32), torch.empty(n_frame, **f32)
        drgb = torch.randn(n_frame, 3, generator=gk, **f32)
        dsg, dcr = torch.empty(n_frame, S, **f32), torch.empty(n_frame, S, 3, **f32)
        wc, zc = torch.rand(n_frame, 64, generator=gk, **f32), zz[:, ::3].contiguous()
        zf = torch.empty(n_frame, 192, **f32)
        ms_cf = timed_kernel(lambda: renderer.k_composite_fwd(sig, None, crw, zz, n_frame, S, w_, rgb_, dep_, acc_))
        ms_cb = timed_kernel(lambda: renderer.k_composite_bwd(sig, None, crw, zz, drgb, n_frame, S, dsg, dcr))
        ms_sp = timed_kernel(lambda: renderer.k_sample_pdf(wc, zc, None, n_frame, None, None, None, zf))
        nsamp = n_frame * S

        def hbm(ms, nbytes, kname):
            g = nbytes / ms / 1e6
            r = {"ms": round(ms, 4), "GB/s": round(g, 1), "frac_of_8TB/s": round(g / 8000.0, 3), "algorithmic_bytes": int(nbytes)}
            r.update(hbm_traffic(kname) if (HW, S) == (400, 192) else {"traffic": None})
            return r
        hbm_kernels = {"workload": f"{n_frame} rays x {S} samples (resampling: 64 coarse -> 192 fine)",
                       "timing": "HIP events on the launch stream, 10 launches back to back; rocprofv3 --kernel-trace of "
                                 "the same shape: profiles/r2_final_hbm_kernel_stats.csv (scripts/hbm_prof.sh)",
                       "composite_fwd": hbm(ms_cf, 24.0 * nsamp + 20.0 * n_frame, "composite_fwd_k<3>"),
                       "composite_bwd": hbm(ms_cb, 40.0 * nsamp + 12.0 * n_frame, "composite_bwd_k<3>"),
                       "sample_pdf": hbm(ms_sp, n_frame * (8.0 * 64 + 4.0 * 192), "sample_pdf_kernel")}
        del sig, crw, zz, w_, dsg, dcr

        cpu = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(base, t_o, t_d, t_b, t_l, args.cpu_seconds)

    except _ExtrasAtN1Only:
        extras_skipped = "the extra single-GPU legs run at N = 1 only (--extras-at-any-n runs them on every rank)"
    except Exception as e:   # noqa: BLE001
        if world > 1:
            raise
        extras_error = repr(e)

    if rank == 0:
        line = {"metric": "rays/sec (train step)", "value": round(train_rays_s, 1), "unit": "rays/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                "data": "synthetic",
                "config": {"workload": "full NeRF coarse+fine 64+128 samples/ray, 8x256 MLP, synthetic 400x400 "
                                       "Blender-style scene; train step = fwd+L2+bwd+Adam",
                           "rays_per_gpu_per_step": B, "global_batch": B * world, "samples_per_ray": 192,
                           "mlp_evals_per_ray": 256, "parallelism": f"dp{world}",
                           "exchange": ("native ncclAllReduce (nerfhip_dp_*)" if native else "torch.distributed all_reduce (RCCL)") if dp else "none"},
                # samples/s counts network evaluations: 64 coarse + 192 fine per ray (the 192 of samples_per_ray are the
                # compositing samples of the fine pass)
                "samples_per_s": round(train_rays_s * 256, 1),
                "roofline": roofline, "render": render, "render_sharded_800": render_sharded, "render_bf16": render_bf16, "render_f16x3": render_f16x3,
                "train_f16x3": train_f16x3, "train_bf16": train_bf16,
                "config5_bf16": config5, "render_pose_800": render_pose, "hbm_kernels": hbm_kernels,
                "cpu_baseline": cpu, "extras_error": extras_error, "extras_skipped": extras_skipped}
        print(json.dumps(line))
        sys.stdout.flush()

    # ---- behind the headline line (it is out and flushed): the library's OWN exchange (nerfhip_dp_train_step_dev:
    # backward + ncclAllReduce + Adam(1/world) in one C call, no torch in the step) beside the torch path, from the same
    # start state on the same batch.  Reported on stderr; a failure or a stall here ends the run non-zero, the line above
    # stays valid.  (runNerf.scala:95-106: what DJL's trainer.step() does with several devices.)
    rc_native = 0
    if dp and not args.no_native_leg:
        def gave_up():
            print(f"[bench native_rccl] rank {rank}: no result within {args.native_leg_timeout:.0f} s -- giving up "
                  "(the headline line above is complete)", file=sys.stderr, flush=True)
            os._exit(3)
        dog = threading.Timer(args.native_leg_timeout, gave_up)
        dog.daemon = True
        dog.start()
        try:
            nsteps = 10
            a = trainer
            a.init_params(7)                       # start state again (also clears Adam's moments and update count)
            b_ = NerfBlock(NerfConfig.from_dict(dict(base, maxRays=B)))
            b_.set_stream(stream)
            b_.init_params(7)
            native_init(b_)

            def step_torch():
                a.train_backward_dev(t_o, t_d, t_b, t_d, t_l, B)
                dist.all_reduce(grad)
                a.optimizer_step(1.0 / world)

            def step_native():
                b_.dp_train_step_dev(t_o, t_d, t_b, t_d, t_l, B)

            def timed(fn):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize(); barrier()
                t0_ = time.perf_counter()
                for _ in range(nsteps):
                    fn()
                torch.cuda.synchronize(); barrier()
                d_ = torch.tensor([time.perf_counter() - t0_], device="cuda", dtype=torch.float64)
                dist.all_reduce(d_, op=dist.ReduceOp.MAX)
                return 1e3 * float(d_.item()) / nsteps
            ms_t = timed(step_torch)
            ms_n = timed(step_native)
            pa, pb = a.get_params(), b_.get_params()
            diff = float(np.abs(pa - pb).max()) if np.isfinite(pa).all() and np.isfinite(pb).all() else float("nan")
            dmax = torch.tensor([diff if diff == diff else 1e30], device="cuda", dtype=torch.float64)
            dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
            diff = float(dmax.item())
            r_, w_ = b_.dp_world()
            ok = diff < 1e-3 and (r_, w_) == (rank, world)
            if rank == 0:
                print("[bench native_rccl] " + json.dumps({"n_gpus": world, "steps": nsteps, "ms_per_step": round(ms_n, 3),
                      "torch_path_ms_per_step": round(ms_t, 3), "rays_per_s": round(world * B / ms_n * 1e3, 1),
                      "max_abs_param_diff_vs_torch": diff, "ok": ok}), file=sys.stderr, flush=True)
            b_.close()
            if not ok:
                rc_native = 3
        except Exception as e:   # noqa: BLE001
            print(f"[bench native_rccl] rank {rank}: FAILED: {e!r}", file=sys.stderr, flush=True)
            rc_native = 3
        dog.cancel()
    if rc_native:
        sys.stderr.flush()
        os._exit(rc_native)      # the other ranks may be parked in a collective: no orderly teardown to wait for
    if dp:
        dist.destroy_process_group()


def _pmc_set(tag):
    """prefix (profiles/<round>_<tag>) of the committed PMC summaries that were collected on THIS library: the meta file
    scripts/pmc_train.sh / hbm_prof.sh write beside them carries nerfhip_version() -- release + hash of the kernel sources
    -- and a set whose string differs from the loaded library's is stale (a kernel changed since) and not used."""
    import glob
    from nerf_amd import _lib
    mine = _lib.load().nerfhip_version().decode()
    seen = []
    for meta in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{tag}_pmc_meta.json")), reverse=True):
        try:
            ver = json.load(open(meta)).get("nerfhip_version")
        except (OSError, ValueError):
            continue
        if ver == mine:
            return meta[:-len("_pmc_meta.json")], None
        seen.append(os.path.basename(meta))
    return None, (f"no PMC summary of this library ({mine}) under profiles/: " +
                  (f"{', '.join(seen)} were collected on other sources" if seen else f"no *_{tag}_pmc_meta.json"))


def pmc_traffic(dom, dtype):
    """HBM bytes per train step of the dominant kernel from the committed rocprofv3 PMC passes of this workload
    (scripts/pmc_train.sh: separate FETCH_SIZE and WRITE_SIZE passes over 3 steps of the same 4096-ray batch;
    units KB; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect
    counters itself, so this is read from profiles/; null -- with the reason -- when the summaries are absent or were
    collected on a library built from other kernel sources."""
    import csv
    sub = {"mlp_wgrad": "wgrad_%s_kernel", "mlp_fwd": "mlp_fwd_%s_kernel<8, true>" if dtype == "f32" else "mlp_fwd_%s_kernel<8, 1, true>",
           "mlp_bwd": "mlp_bwd_%s_kernel"}[dom] % dtype
    tag = "f16x3train" if dtype == "f16x3" else dtype
    prefix, why = _pmc_set(tag)
    if prefix is None:
        return {"traffic": None, "traffic_note": why}
    tot = 0.0
    try:
        for kind, col, mul in (("fetchsize", "FETCH_SIZE", 2.0), ("writesize", "WRITE_SIZE", 1.0)):
            for r in csv.DictReader(open(f"{prefix}_pmc_{kind}.csv")):
                if sub in r["kernel"]:
                    tot += mul * float(r[col]) * 1024.0 / 3.0     # 3 profiled steps
    except (OSError, KeyError, ValueError) as e:
        return {"traffic": None, "traffic_note": repr(e)}
    rel = os.path.relpath(prefix, ROOT)
    return {"traffic": round(tot), "traffic_unit": "bytes per train step (all launches of the kernel)",
            "traffic_source": f"{rel}_pmc_fetchsize.csv, _writesize.csv"}


def hbm_traffic(kname):
    """HBM bytes per launch of one of the per-ray kernels at the 160 000 x 192 shape, from the committed PMC passes
    (scripts/hbm_prof.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KB, 6 launches each; FETCH_SIZE doubled per
    the gfx950 correction of MI355X_MICROARCH.md -- for these one-dword-to-four-dword-per-lane loads the doubled figure
    reproduces the algorithmic read bytes to 2 %); null when the summaries belong to other kernel sources."""
    import csv
    prefix, why = _pmc_set("hbm")
    if prefix is None:
        return {"traffic": None, "traffic_note": why}
    tot = 0.0
    try:
        for kind, col, mul in (("fetchsize", "FETCH_SIZE", 2.0), ("writesize", "WRITE_SIZE", 1.0)):
            for r in csv.DictReader(open(f"{prefix}_pmc_{kind}.csv")):
                if kname in r["kernel"]:
                    tot += mul * float(r[col]) * 1024.0 / float(r["dispatches"])
    except (OSError, KeyError, ValueError, ZeroDivisionError) as e:
        return {"traffic": None, "traffic_note": repr(e)}
    rel = os.path.relpath(prefix, ROOT)
    return {"traffic": round(tot), "traffic_source": f"{rel}_pmc_fetchsize.csv, _writesize.csv"} if tot else {"traffic": None}


FLOP_TRAIN_RAY = 789e6     # forward + backward of one ray, 64 coarse + 192 fine evaluations of the full network
FLOP_RENDER_RAY = 264.3e6  # eval render of one ray


def _median_rate(fn, n_target, seconds, step=64):
    """rays/s of fn(n) (n = rays per call) by SURVEY 8d's protocol -- warm-up 3, median of 10 calls -- bounded by a time
    budget: a probe on `step` rays estimates the time of a call; n = n_target unless even 1 warm-up + 2 timed calls of
    that size would overrun the budget (then the largest multiple of `step` that fits), and fewer than 3 + 10 calls are made
    when they do not fit.  Returns (rays/s, n, warm-up calls, timed calls)."""
    fn(step)
    t0 = time.perf_counter(); fn(step); per_ray = (time.perf_counter() - t0) / step
    n = n_target
    if 3 * per_ray * n > seconds:
        n = int(max(step, min(n_target, seconds / (3 * per_ray)) // step * step))
    est = per_ray * n
    reps = int(max(2, min(10, seconds / est - 1)))
    warm = int(max(1, min(3, seconds / est - reps)))
    for _ in range(warm):
        fn(n)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(n); ts.append(time.perf_counter() - t0)
    return n / float(np.median(ts)), n, warm, reps


def cpu_baseline(base, t_o, t_d, t_b, t_l, seconds):
    """The reference's CPU path timed on this box's host cores, twice, on a bounded sample of the same workload.

    The reference on a CPU is DJL -> libtorch CPU operators (build.sbt:10-13; runNerf.scala:94-106).  It cannot run here (no
    JVM), so two stand-ins are timed:  "torch_cpu" -- the op-by-op transliteration of nerfBlock.scala / nerfCoreBlock.scala
    onto torch CPU tensors (tests/golden/make_golden.py: the same libtorch operator families DJL dispatches to, autograd
    backward, DJL-form Adam), and "oracle" -- this repository's C++ restatement (oracle/, OpenMP + a hand-blocked sgemm).
    Protocol: a 1024-ray train step of the SAME batch and network, a 1024-ray eval render of it, and BASELINE
    configs[0] (the reference's own CPU-runnable case: 100x100 frame = 10 000 rays, 64 samples, no fine pass, small MLP);
    warm-up 3 / median of 10 where the budget allows, the counts used are reported.  `value` = the FASTER of the two on the
    train step, so GPU/CPU ratios are quoted against the stronger baseline."""
    import torch
    nthreads = int(os.environ.get("OMP_NUM_THREADS", "0")) or torch.get_num_threads()
    from oracle.oracle import Oracle
    import helpers as H
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as MG                      # sets torch's thread count to 4 on import (fixture generation) ...
    torch.set_num_threads(nthreads)               # ... the baseline uses the box's CPU share
    c = {k: v for k, v in base.items() if k not in ("device", "dtype", "seed", "maxRays")}
    c0 = H.cfg(D=4, W=64, skips=[2], posL=6, dirL=2, NSamples=64, NImportance=0, useHierarchical=False)   # configs[0]
    nb = 1024
    ro, rd, bd, lb = (x[:nb].cpu().numpy() for x in (t_o, t_d, t_b, t_l))
    params = H.make_params(c, 3, bias_scale=0.0)
    params0 = H.make_params(c0, 5)
    o_full, o_tiny = Oracle(c), Oracle(c0)
    hw0 = 100
    f0 = 0.5 * hw0 / math.tan(0.5 * 0.6911112070083618)
    pose0 = np.array([[1, 0, 0, 0], [0, 0.5, -0.8660254, -3.4641016], [0, 0.8660254, 0.5, 2.0]], np.float32)
    r0o, r0d, r0b = o_tiny.gen_rays(hw0, hw0, f0, pose0, 2.0, 6.0, False)
    # ---- torch CPU
    T = torch.from_numpy
    tro, trd, tbd, tlb = T(ro), T(rd), T(bd), T(lb)
    w = T(params.copy()).requires_grad_(True)
    m_ = torch.zeros_like(w); v_ = torch.zeros_like(w)
    state = {"t": 0}
    factor = float(np.float32(0.1 ** (1.0 / (c["lrateDecay"] * 1000.0))))

    def torch_train(n):
        nets = H.split_params(c, w)
        out = MG.forward(c, nets, tro[:n], trd[:n], tbd[:n], trd[:n], True, {})
        loss = ((tlb[:n] - out["rgb_fine"]) ** 2).mean() + ((tlb[:n] - out["rgb_coarse"]) ** 2).mean()     # runNerf.scala:72-73
        w.grad = None
        loss.backward()
        with torch.no_grad():     # DJL Adam: bias correction folded into the step size, eps outside the root
            state["t"] += 1; t = state["t"]
            lr = max(c["lrate"] * factor ** t, 1e-8)
            lrp = lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
            m_.mul_(0.9).add_(w.grad, alpha=0.1)
            v_.mul_(0.999).addcmul_(w.grad, w.grad, value=0.001)
            w.sub_(lrp * m_ / (v_.sqrt() + 1e-7))

    def torch_render(n):
        with torch.no_grad():
            MG.forward(c, H.split_params(c, w.detach()), tro[:n], trd[:n], tbd[:n], trd[:n], False, {})
    w0 = T(params0.copy())
    t0o, t0d, t0b = T(r0o), T(r0d), T(r0b)

    def torch_c1(n):
        with torch.no_grad():
            MG.forward(c0, H.split_params(c0, w0), t0o[:n], t0d[:n], t0b[:n], t0d[:n], False, {})
    n0 = hw0 * hw0
    tt = _median_rate(torch_train, nb, seconds * 0.30)
    tr = _median_rate(torch_render, nb, seconds * 0.12)
    tc = _median_rate(torch_c1, n0, seconds * 0.10, step=1000)

    # ---- C++ restatement
    pp = params.copy(); m = np.zeros_like(pp); v = np.zeros_like(pp)
    ostate = {"t": 0}

    def port_train(n):
        r = o_full.train_grads(pp, ro[:n], rd[:n], bd[:n], rd[:n], lb[:n])
        ostate["t"] += 1
        o_full.adam(pp, r["grads"], m, v, ostate["t"])
    ot = _median_rate(port_train, nb, seconds * 0.26)
    orr = _median_rate(lambda n: o_full.forward(pp, ro[:n], rd[:n], bd[:n], rd[:n]), nb, seconds * 0.12)
    oc = _median_rate(lambda n: o_tiny.forward(params0, r0o[:n], r0d[:n], r0b[:n], r0d[:n]), n0, seconds * 0.10, step=1000)

    def leg(tr_, re_, c1_):
        return {"train_rays_per_s": round(tr_[0], 2), "train_gflops": round(tr_[0] * FLOP_TRAIN_RAY / 1e9, 1),
                "render_rays_per_s": round(re_[0], 2), "render_gflops": round(re_[0] * FLOP_RENDER_RAY / 1e9, 1),
                "c1_render_rays_per_s": round(c1_[0], 1),
                "rays_per_call__warmup_calls__timed_calls": {"train": list(tr_[1:]), "render": list(re_[1:]), "c1": list(c1_[1:])}}
    torch_leg = leg(tt, tr, tc)
    port_leg = leg(ot, orr, oc)
    best = "torch_cpu" if torch_leg["train_rays_per_s"] >= port_leg["train_rays_per_s"] else "oracle"
    bleg = torch_leg if best == "torch_cpu" else port_leg
    return {"value": bleg["train_rays_per_s"], "unit": "rays/s", "cores": nthreads, "kind": "port", "impl": best,
            "render_value": max(torch_leg["render_rays_per_s"], port_leg["render_rays_per_s"]),
            "sample": f"median time of a train step (fwd+loss+bwd+Adam) on the first {nb} rays of the same batch and network "
                      f"(fewer where the time budget forces it: rays per call, warm-up and timed calls are listed per leg); render: "
                      f"the same rays in eval mode; c1: BASELINE configs[0], one {hw0}x{hw0} frame ({hw0 * hw0} rays, 64 samples, D=4 W=64)",
            "torch_cpu": dict(torch_leg, threads=nthreads, torch=torch.__version__,
                              what="tests/golden/make_golden.py forward + autograd + DJL-form Adam on CPU tensors"),
            "oracle": dict(port_leg, threads=o_full.num_threads(), what="oracle/nerf_oracle.cpp (OpenMP, blocked sgemm)"),
            "note": "neither is the reference itself (Scala/DJL cannot run here: no JVM); torch_cpu uses the libtorch CPU "
                    "operator families DJL dispatches to, value/render_value are the faster of the two"}


if __name__ == "__main__":
    main()
