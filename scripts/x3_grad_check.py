"""Gradient accuracy of the dtype-f16x3 training step (split-f16 forward and backward-data, fp32 weight gradient) next
to the all-fp32 step, both against the CPU oracle: full network, 256 rays."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from nerf_amd import NerfBlock, NerfConfig
from oracle.oracle import Oracle
c = H.CONFIGS["full"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ro, rd, b, label = H.make_rays(n, 21)
params = H.make_params(c, 2)
ref = Oracle(c).train_grads(params, ro, rd, b, rd, label)["grads"]
out = {}
for dt in ("f32", "f16x3"):
    blk = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n, dtype=dt)))
    blk.set_params(params)
    lf, lc = blk.train_backward(ro, rd, b, rd, label)
    out[dt] = blk.get_grads()
    blk.close()
    g = out[dt]
    print(f"{dt:6s} loss {lf:.7f} {lc:.7f}  rel L2 vs oracle {np.linalg.norm(g - ref) / np.linalg.norm(ref):.3e}  max abs / max {np.abs(g - ref).max() / np.abs(ref).max():.3e}")
print(f"f16x3 vs f32: rel L2 {np.linalg.norm(out['f16x3'] - out['f32']) / np.linalg.norm(out['f32']):.3e}")
off = 0
for net in range(2):
    for nm, r, k in H.layout(c):
        a, e, f = out["f16x3"][off:off + r * k], ref[off:off + r * k], out["f32"][off:off + r * k]
        off += r * k
        print(f"  net {net} {nm:8s} x3 {np.linalg.norm(a - e) / max(np.linalg.norm(e), 1e-30):.2e}   f32 {np.linalg.norm(f - e) / max(np.linalg.norm(e), 1e-30):.2e}")
