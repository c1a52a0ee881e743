"""Per-kernel time of a training step (full network, 4096 rays) from the library's own HIP-event profile:
python scripts/train_time.py <dtype> [steps] [library]  -- the optional library path is for development builds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import time
import numpy as np, torch
import helpers as H
from nerf_amd import _lib
if len(sys.argv) > 3:
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
from nerf_amd import NerfBlock, NerfConfig
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = dict(H.CONFIGS["full"], dtype=dtype)
nT = 4096
tr = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nT)))
tr.init_params(1)
ro, rd, b, lab = H.make_rays(nT, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
torch.cuda.synchronize()
for _ in range(3):
    tr.train_step_dev(o, d, bb, d, l, nT)
tr.synchronize()
tr.set_profiling(True)
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step_dev(o, d, bb, d, l, nT)
tr.synchronize()
dt = time.perf_counter() - t0
pr = tr.get_profile()
print(f"{dtype} {os.path.basename(_lib.LIB_PATH)}: {1e3 * dt / steps:.3f} ms/step  " +
      "  ".join(f"{k} {pr[k + '_ms'] / steps:.3f}" for k in ("mlp_fwd", "mlp_bwd", "mlp_wgrad")))
