"""Diagnostic (library built by scripts/x3_stamps.sh with -DNERFHIP_DIAG_STAMPS=1): cycle stamps of one wave's 4th tile of
the split-f16 TRAINING forward (fine network of a 4096-ray step: activation / sign-bit stores among the MFMAs), next to
the inference forward of the same samples.  python scripts/x3f_train_stamps.py build/libnerfhip_stamps.so"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig, _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
c = dict(H.CONFIGS["full"], dtype="f16x3")
n = 4096
blk = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
blk.init_params(1)
ro, rd, b, lab = H.make_rays(n, 1)
names = ["encode", "layer0"] + [f"layer{i}" for i in range(1, 8)] + ["(gap)", "head", "colour"]
for _ in range(2):
    lf, lc, dbg = blk.train_backward(ro, rd, b, rd, lab, debug=True)      # stamps land in the first bytes of craw_f
st_t = dbg["craw_f"].reshape(-1)[:26].view(np.uint64).astype(np.int64)
fine, coarse, dbg = blk.forward(ro, rd, b, rd, training=False, debug=True)
st_i = dbg["craw_f"].reshape(-1)[:26].view(np.uint64).astype(np.int64)
dt, di = np.diff(st_t), np.diff(st_i)
print(f"{'part':8s} {'train':>8s} {'infer':>8s} {'MFMA':>8s}   (cycles of one wave; MFMA = 3 x k-steps x tiles x 32)")
mf = [0, 3 * 4 * 8 * 32] + [3 * (20 if i == 5 else 16) * 8 * 32 for i in range(1, 8)] + [0, 3 * 16 * 5 * 32, 3 * 8 * 32]
for nm, a, b_, m in zip(names, dt, di, mf):
    print(f"{nm:8s} {a:8d} {b_:8d} {m:8d}")
print("tile total", st_t[12] - st_t[0], st_i[12] - st_i[0], sum(mf))
