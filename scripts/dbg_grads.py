import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle.oracle import Oracle
from nerf_amd import NerfBlock, NerfConfig
case = sys.argv[1] if len(sys.argv) > 1 else "small_dir"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
c = H.CONFIGS[case]
g = np.load(os.path.join(ROOT, "tests/golden", case + "_train.npz"))
params = H.make_params(c, int(g["seed"]))
blk = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=int(g["rays_o"].shape[0]), dtype=dtype)))
blk.set_params(params)
rand = {k[5:]: g[k] for k in g.files if k.startswith("rand_")}
lf, lc = blk.train_backward(g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], g["label"], rand=rand or None)
gr = blk.get_grads()
rg = Oracle(c).train_grads(params, g["rays_o"], g["rays_d"], g["bounds"], g["rays_d"], g["label"], rand=rand or None)["grads"]
names = H.layout(c)
off = 0
nets = 2 if c["useHierarchical"] else 1
for n in range(nets):
    for nm, r_, c_ in names:
        shp = (r_, c_); k = r_ * c_
        a, b = gr[off:off + k], rg[off:off + k]
        print(n, nm, shp, "relerr %.3e" % (np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)), "norm %.3e" % np.linalg.norm(b))
        off += k
