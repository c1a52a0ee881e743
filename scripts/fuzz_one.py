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


def poscode(x, L):   # nerfCoreBlock.scala:127-136
    out = [x]
    for i in range(L):
        y = x * float(1 << i)
        out += [np.sin(y), np.cos(y)]
    return np.concatenate(out, -1)


def fence_report(net, nm, row, z):
    """the unit's pre-activation at every sample, in float64 from the same parameters, rays and depths: a ReLU on the fence
    has one sample within fp32 rounding of zero; a wrong mask bit would show no such sample"""
    p = {k: v.astype(np.float64) for k, v in H.split_params(c, params)[net].items()}
    pos = (ro[:, None, :] + rd[:, None, :] * z[:, :, None]).astype(np.float64)   # fp32 products and sums, as the device
    enc = poscode(pos, c["posL"])
    h = enc
    pre = None
    for i in range(c["D"]):
        a = h @ p[f"w{i}"].T + p[f"b{i}"].reshape(-1)
        if nm == f"w{i}":
            pre = a[..., row]
        h = np.maximum(a, 0)
        if i in c["skips"]:
            h = np.concatenate([h, enc], -1)
    if nm == "ws":
        pre = (h @ p["ws"].T + p["bs"].reshape(-1))[:, :-1, 0]
        key = "noise_f" if (net == 1 or not c["useHierarchical"]) else "noise_c"
        if key in rand:
            pre = pre + rand[key][:, :pre.shape[1]].astype(np.float64) * c["rawNoiseStd"]
    if nm == "wf":   # colour hidden layer: relu(feature part + direction part), nerfCoreBlock.scala:96-104
        a = h @ p["wf"].T + p["bf"].reshape(-1)
        b = poscode(rd.astype(np.float64), c["dirL"]) @ p["wd"].T + p["bd"].reshape(-1)
        pre = (a + b[:, None, :])[..., row]
    ap = np.abs(pre).ravel()
    k = np.argsort(ap)[:3]
    print("      pre-activation of that unit over the %d samples (float64): smallest |.| %s at (ray, sample) %s; median |.| %.3g"
          % (ap.size, ["%.2e" % pre.ravel()[j] for j in k], [tuple(int(t) for t in np.unravel_index(j, pre.shape)) for j in k], np.median(ap)))


for dtype in (("f32", "f16x3") if c["W"] >= 64 else ("f32",)):
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
                if nm != "wd":
                    fence_report(net, nm, int(top[0]), dbg["z_fine"] if (net == 1 and c["useHierarchical"]) else dbg["z_coarse"])
        if nm == "ws" and e > 2e-3 * nn:
            fence_report(net, nm, 0, dbg["z_fine"] if (net == 1 and c["useHierarchical"]) else dbg["z_coarse"])
    blk.close()
