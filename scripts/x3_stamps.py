"""Diagnostic (library built by scripts/x3_stamps.sh with -DNERFHIP_DIAG_STAMPS=1): cycle stamps of one wave's 4th tile of the split-f16 forward."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig, _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
c = dict(H.CONFIGS["full"], dtype="f16x3")
n, S = 160000, 192
blk = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
blk.init_params(1)
ro, rd, b, lab = H.make_rays(n, 1)
dev = lambda a: torch.from_numpy(a).cuda()
z = torch.sort(torch.rand(n, S, device="cuda") * 4 + 2, dim=1).values.contiguous()
sig, craw = torch.empty(n, S, device="cuda"), torch.empty(n, S, 3, device="cuda")
for _ in range(3):
    blk.k_mlp_forward(1, dev(ro), dev(rd), dev(rd), z, n, S, sig, craw)
blk.synchronize()
st = craw.view(-1)[:26].cpu().numpy().view(np.uint64).astype(np.int64)
names = ["encode", "layer0"] + [f"layer{i}" for i in range(1, 8)] + ["(gap)", "head", "colour"]
d = np.diff(st)
for nm, v in zip(names, d): print(f"{nm:8s} {v:8d}")
print("tile total", st[12] - st[0])
