#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native NeRF hot path (libnerfhip.so).

Workload (BASELINE.json configs[1] / configs[3]): full NeRF coarse+fine (64+128 samples/ray), 8x256 MLP with
skip at 4, posL 10 / dirL 4 view-direction head, synthetic Blender-style 400x400 scene, fp32 MFMA.
  * a "step" = ONE training iteration (forward + L2 loss + backward + [grad all-reduce] + Adam) on
    4096 rays per GPU (weak scaling; runNerf.scala:94-106), inputs already resident in HBM;
  * after the timed train steps the same process renders 400x400 frames (160 000 rays each) and reports
    render rays/s in the "render" object of the same JSON line.
`value` is whole-job train-step rays/s (all ranks).  Launch: python bench.py --gpus 1, or under
torch.distributed.run for N > 1 (one process per GPU, RCCL all-reduce of the flat gradient buffer).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16x3": 833.3}   # MI355X_MICROARCH.md: dense MFMA peaks (f16x3: the f16 peak / 3 MFMAs per product)


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
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--native-rccl", action="store_true", help="data-parallel exchange through the library's own RCCL "
                    "entry points (nerfhip_dp_*: no torch in the step) instead of torch.distributed.all_reduce on the "
                    "gradient buffer; torch.distributed still provides rendezvous, barriers and the max-over-ranks time")
    ap.add_argument("--extras-at-any-n", action="store_true", help="run the extra single-GPU legs (other dtypes, config 5, "
                    "800x800 frames, HBM kernels) on every rank at N > 1 as well; by default they run at N = 1 only")
    ap.add_argument("--force-dp", action="store_true", help="take the data-parallel code path (backward, all-reduce, "
                    "optimizer step) even with one rank: rehearsal of the N > 1 loop on a 1-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nerf_amd import NerfBlock, NerfConfig
    from nerf_amd.dp import grad_tensor

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
                "roofline": roofline, "render": render, "render_bf16": render_bf16, "render_f16x3": render_f16x3,
                "train_f16x3": train_f16x3, "train_bf16": train_bf16,
                "config5_bf16": config5, "render_pose_800": render_pose, "hbm_kernels": hbm_kernels,
                "cpu_baseline": cpu, "extras_error": extras_error, "extras_skipped": extras_skipped}
        print(json.dumps(line))
    if dp:
        dist.destroy_process_group()


def pmc_traffic(dom, dtype):
    """HBM bytes per train step of the dominant kernel from the committed rocprofv3 PMC passes of this workload
    (scripts/pmc_train.sh: separate FETCH_SIZE and WRITE_SIZE passes over 3 steps of the same 4096-ray batch;
    units KB; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect
    counters itself, so this is read from profiles/; null when the summaries are absent."""
    import csv
    sub = {"mlp_wgrad": "wgrad_%s_kernel", "mlp_fwd": "mlp_fwd_%s_kernel<8, true>" if dtype == "f32" else "mlp_fwd_%s_kernel<8, 1, true>",
           "mlp_bwd": "mlp_bwd_%s_kernel"}[dom] % dtype
    tag = "f16x3train" if dtype == "f16x3" else dtype
    tot = 0.0
    try:
        for kind, col, mul in (("fetchsize", "FETCH_SIZE", 2.0), ("writesize", "WRITE_SIZE", 1.0)):
            path = os.path.join(ROOT, "profiles", f"r2_final_{tag}_pmc_{kind}.csv")
            for r in csv.DictReader(open(path)):
                if sub in r["kernel"]:
                    tot += mul * float(r[col]) * 1024.0 / 3.0     # 3 profiled steps
    except (OSError, KeyError, ValueError):
        return {"traffic": None}
    return {"traffic": round(tot), "traffic_unit": "bytes per train step (all launches of the kernel)",
            "traffic_source": f"profiles/r2_final_{tag}_pmc_fetchsize.csv, _writesize.csv"}


def hbm_traffic(kname):
    """HBM bytes per launch of one of the per-ray kernels at the 160 000 x 192 shape, from the committed PMC passes
    (scripts/hbm_prof.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KB, 6 launches each; FETCH_SIZE doubled per
    the gfx950 correction of MI355X_MICROARCH.md -- for these one-dword-to-four-dword-per-lane loads the doubled figure
    reproduces the algorithmic read bytes to 2 %)."""
    import csv
    tot = 0.0
    try:
        for kind, col, mul in (("fetchsize", "FETCH_SIZE", 2.0), ("writesize", "WRITE_SIZE", 1.0)):
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"r2_final_hbm_pmc_{kind}.csv"))):
                if kname in r["kernel"]:
                    tot += mul * float(r[col]) * 1024.0 / float(r["dispatches"])
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return {"traffic": None}
    return {"traffic": round(tot), "traffic_source": "profiles/r2_final_hbm_pmc_fetchsize.csv, _writesize.csv"} if tot else {"traffic": None}


def cpu_baseline(base, t_o, t_d, t_b, t_l, seconds):
    """The CPU restatement (oracle, kind "port") timed on this box's host cores on a bounded sample of the
    same workload: train iterations on 256-ray slices of the same batch, then eval renders."""
    from oracle.oracle import Oracle
    import helpers as H
    c = {k: v for k, v in base.items() if k not in ("device", "dtype", "seed", "maxRays")}
    o = Oracle(c)
    params = H.make_params(c, 3, bias_scale=0.0)
    ro, rd, bd, lb = (x.cpu().numpy() for x in (t_o, t_d, t_b, t_l))
    chunk = 256
    m = np.zeros_like(params); v = np.zeros_like(params)
    o.train_grads(params, ro[:32], rd[:32], bd[:32], rd[:32], lb[:32])   # warm-up
    t0 = time.perf_counter(); done = 0; it = 0
    while time.perf_counter() - t0 < seconds * 0.6:
        s = (it * chunk) % (ro.shape[0] - chunk + 1)
        r = o.train_grads(params, ro[s:s + chunk], rd[s:s + chunk], bd[s:s + chunk], rd[s:s + chunk], lb[s:s + chunk])
        o.adam(params, r["grads"], m, v, it + 1)
        done += chunk; it += 1
    train = done / (time.perf_counter() - t0)
    t0 = time.perf_counter(); rdone = 0; it = 0
    while time.perf_counter() - t0 < seconds * 0.4:
        s = (it * chunk) % (ro.shape[0] - chunk + 1)
        o.forward(params, ro[s:s + chunk], rd[s:s + chunk], bd[s:s + chunk], rd[s:s + chunk])
        rdone += chunk; it += 1
    rend = rdone / (time.perf_counter() - t0)
    return {"value": round(train, 2), "unit": "rays/s", "cores": o.num_threads(), "kind": "port",
            "sample": f"{done} rays of the same 4096-ray batch in 256-ray train iterations (fwd+loss+bwd+Adam); "
                      f"render: {rdone} rays", "render_value": round(rend, 2),
            "note": "C++ restatement of the reference path (oracle/), OpenMP; the Scala/DJL reference cannot run here"}


if __name__ == "__main__":
    main()
