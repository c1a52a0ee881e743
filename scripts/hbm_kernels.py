"""The HBM-bound per-ray kernels alone, at the shape of one 400x400 frame's fine pass (160 000 rays x 192 samples;
resampling 64 coarse -> 192 fine), for rocprofv3 (kernel trace and PMC passes: scripts/hbm_prof.sh) and for quick
timing.  usage: python3 scripts/hbm_kernels.py [reps]   -> one JSON line with HIP-event times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
from nerf_amd import NerfBlock, NerfConfig

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, S, N = 160000, 192, 64
blk = NerfBlock(NerfConfig(maxRays=64))
s = torch.cuda.Stream(); torch.cuda.set_stream(s); blk.set_stream(s.cuda_stream)
f32 = dict(dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(5)
sig = torch.rand(n, S, generator=g, **f32) * 3.0
crw = torch.randn(n, S, 3, generator=g, **f32)
zz = torch.sort(torch.rand(n, S, generator=g, **f32) * 4.0 + 2.0, dim=1).values.contiguous()
w_, rgb_, dep_, acc_ = torch.empty(n, S, **f32), torch.empty(n, 3, **f32), torch.empty(n, **f32), torch.empty(n, **f32)
drgb = torch.randn(n, 3, generator=g, **f32)
dsg, dcr = torch.empty(n, S, **f32), torch.empty(n, S, 3, **f32)
wc, zc = torch.rand(n, N, generator=g, **f32), zz[:, ::3].contiguous()
zf = torch.empty(n, S, **f32)
bd = torch.stack([torch.full((n,), 2.0, **f32), torch.full((n,), 6.0, **f32)], 1).contiguous()
zs = torch.empty(n, N, **f32)
kern = {
    "composite_fwd": (lambda: blk.k_composite_fwd(sig, None, crw, zz, n, S, w_, rgb_, dep_, acc_), 24.0 * n * S + 20.0 * n),
    "composite_bwd": (lambda: blk.k_composite_bwd(sig, None, crw, zz, drgb, n, S, dsg, dcr), 40.0 * n * S + 12.0 * n),
    "sample_pdf": (lambda: blk.k_sample_pdf(wc, zc, None, n, None, None, None, zf), n * (8.0 * N + 4.0 * S)),
    "stratified": (lambda: blk.k_stratified(bd, None, n, zs), n * (8.0 + 4.0 * N)),
}
out = {}
for name, (fn, nbytes) in kern.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "algorithmic_bytes": int(nbytes)}
print(json.dumps(out))
blk.close()
