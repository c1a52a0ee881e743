"""one-off: one seed of tests/test_gpu_fuzz.py's generator, with the per-tensor gradient errors of every dtype and where
the worst tensor's error sits (rows of the matrix): a single hot row = one ReLU that fell on the other side of zero"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, gpu_util as G
from oracle.oracle import Oracle
import test_gpu_fuzz as F
seed = int(sys.argv[1])
c, n, r = F.draw(seed)
print(c, n)
N, NI = c["NSamples"], c["NImportance"]
S = N + NI if c["useHierarchical"] else N
ro, rd, b, label = H.make_rays(n, 200 + seed)
params = H.make_params(c, 50 + seed)
rand = {}
if c["perturb"]:
    rand["t_rand"] = H.det_uniform(n * N, seed + 7).reshape(n, N)
    if c["useHierarchical"]:
        rand["u_rand"] = H.det_uniform(n * NI, seed + 8).reshape(n, NI)
if c["rawNoiseStd"] > 0:
    rand["noise_c"] = H.det_normal(n * N, seed + 9).reshape(n, N)
    rand["noise_f"] = H.det_normal(n * S, seed + 10).reshape(n, S)
o = Oracle(c)
for dtype in ("f32", "f16x3"):
    blk = G.make_block(c, max_rays=max(n, 8), dtype=dtype)
    blk.set_params(params)
    lf, lc, dbg = blk.train_backward(ro, rd, b, rd, label, rand=rand, debug=True)
    tr = o.train_grads(params, ro, rd, b, rd, label, rand=dict(rand, z_fine_in=dbg["z_fine"]))
    gr, rg = blk.get_grads(), tr["grads"]
    print(dtype, "whole", np.linalg.norm(gr - rg) / np.linalg.norm(rg))
    rows_of = {nm: r for nm, r, _ in H.layout(c)}
    for net, nm, o_, k in H.param_spans(c):
        e = np.linalg.norm(gr[o_:o_ + k] - rg[o_:o_ + k]); nn = np.linalg.norm(rg[o_:o_ + k])
        flag = " <<<" if e > 2e-3 * nn else ""
        print("  ", net, nm, "err %.3e norm %.3e rel %.2e%s" % (e, nn, e / max(nn, 1e-30), flag))
        if e > 1e-4 * nn and nm.startswith("w"):
            rows = rows_of[nm]
            d = (gr[o_:o_ + k] - rg[o_:o_ + k]).reshape(rows, -1)
            rn = np.linalg.norm(d, axis=1)
            top = np.argsort(-rn)[:4]
            print("      rows with most error:", [(int(t), float(rn[t])) for t in top], "all-rows", float(np.linalg.norm(rn)))
            if rows > 1 and rn[top[0]] > 100 * rn[top[1]]:
                print("      => ONE unit: network %d, %s row %d carries %.6f of this tensor's error (the next row %.1e of it)"
                      % (net, nm, top[0], rn[top[0]] / np.linalg.norm(rn), rn[top[1]] / np.linalg.norm(rn)))
    blk.close()
