"""Inference forward rate at training-step sizes: python scripts/render_small.py [f32|bf16] [rays]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = dict(H.CONFIGS["full"], dtype=dtype)
rn = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
rn.init_params(1)
ro, rd, b, lab = H.make_rays(n, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb = dev(ro), dev(rd), dev(b)
out = torch.empty(n, 3, device="cuda")
for _ in range(3):
    rn.render_dev(o, d, bb, d, n, out)
torch.cuda.synchronize()
rn.set_profiling(True)
frames = 20
for _ in range(frames):
    rn.render_dev(o, d, bb, d, n, out)
torch.cuda.synchronize()
p = rn.get_profile()
print(dtype, n, "mlp_fwd ms/frame %.3f TF %.1f" % (p["mlp_fwd_ms"] / frames, p["mlp_fwd_flop"] / p["mlp_fwd_ms"] / 1e9))
