"""Diagnostic: fused forward kernel time vs trunk depth (fixed vs per-layer cost)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig
n, S = 8192, 192
ro, rd, b, _ = H.make_rays(n, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d = dev(ro), dev(rd)
z = dev(np.sort(H.det_uniform(n * S, 5, 2.0, 6.0).reshape(n, S), axis=1))
sig = torch.empty(n, S, device="cuda")
res = {}
for D in (1, 2, 4, 8, 16):
    c = H.cfg(D=D, skips=[], useDir=False, useHierarchical=False, NSamples=S, NImportance=0)
    blk = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
    blk.init_params(1)
    torch.cuda.synchronize()
    blk.k_mlp_forward(0, o, d, d, z, n, S, sig, None); blk.synchronize()
    blk.set_profiling(True)
    for _ in range(3):
        blk.k_mlp_forward(0, o, d, d, z, n, S, sig, None)
    p = blk.get_profile()
    ms = p["mlp_fwd_ms"] / 3
    res[D] = ms
    wgs = n * S / 128
    print(f"D={D:2d} ms={ms:8.3f}  TF={p['mlp_fwd_flop']/p['mlp_fwd_ms']/1e9:7.2f}  us/WG-round={ms*1e3/(wgs/256):8.2f}")
    blk.close()
per_layer = (res[16] - res[8]) / 8
print("per extra layer ms", per_layer, " ideal", 2*256*256*n*S/157.3e12*1e3, " fixed (D=1 minus L0 ideal)", res[1])
