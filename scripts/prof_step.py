"""Small driver for rocprofv3 passes: a few train steps + one 400x400 render of the full fp32 network."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
c = dict(H.CONFIGS["full"], dtype=dtype)
nT, nR = 4096, 40000
tr = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nT)))
rn = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nR)))
tr.init_params(1); rn.set_params(tr.get_params())
ro, rd, b, lab = H.make_rays(nR, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
out = torch.empty(nR, 3, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    tr.train_step_dev(o, d, bb, d, l, nT)
tr.synchronize()
for _ in range(2):
    rn.render_dev(o, d, bb, d, nR, out)
rn.synchronize()
print("done")
