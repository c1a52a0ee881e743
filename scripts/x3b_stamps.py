"""Diagnostic: cycle stamps of one wave's 4th tile in the split-f16 backward-data kernel, from a library built with
-DX3B_STAMPS (python scripts/x3b_stamps.py <library>)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from nerf_amd import NerfBlock, NerfConfig
c = dict(H.CONFIGS["full"], dtype="f16x3")
nT = 4096
tr = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nT)))
tr.init_params(1)
ro, rd, b, lab = H.make_rays(nT, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
for _ in range(3):
    tr.train_step_dev(o, d, bb, d, l, nT)
tr.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
st = np.zeros(32, np.uint64)
assert lib.nerfhip_debug_x3b_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.astype(np.int64)
names = ["prep (loads, scale, head stores)", "Wc^T part", "head part"] + [f"layer {l}" for l in range(7, 0, -1)]
for i, nm in enumerate(names):
    print(f"{nm:34s} {st[i + 1] - st[i]:8d}")
print("tile total", st[len(names)] - st[0])
