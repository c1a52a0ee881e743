"""rocprofv3 driver: a few train steps of the full network: python scripts/prof_train.py [f32|bf16]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
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
print("done")
