"""Render-only timing of the full network (400x400 frames): python scripts/bench_render.py [f32|bf16|f16x3] [frames] [library|-] [NImportance]
(the optional library path is for development builds; NImportance 256 = the sample counts of BASELINE configs[4])"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig, _lib
if len(sys.argv) > 3 and sys.argv[3] != "-":   # an experimental build of the library (development timing only)
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = dict(H.CONFIGS["full"], dtype=dtype)
if len(sys.argv) > 4:
    c["NImportance"] = int(sys.argv[4])
n = 160000
rn = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
rn.init_params(1)
ro, rd, b, lab = H.make_rays(n, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb = dev(ro), dev(rd), dev(b)
out = torch.empty(n, 3, device="cuda")
rn.render_dev(o, d, bb, d, n, out); torch.cuda.synchronize()
rn.set_profiling(True)
t0 = time.perf_counter()
for _ in range(frames):
    rn.render_dev(o, d, bb, d, n, out)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = rn.get_profile()
print(dtype, "rays/s %.0f" % (n * frames / dt), "ms/frame %.2f" % (1e3 * dt / frames),
      "mlp_fwd ms/frame %.2f TF %.1f" % (p["mlp_fwd_ms"] / frames, p["mlp_fwd_flop"] / p["mlp_fwd_ms"] / 1e9),
      {k: round(v / frames, 3) for k, v in p.items() if k.endswith("_ms")})
