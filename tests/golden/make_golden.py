"""Generates tests/golden/*.npz: golden input/output vectors for the NeRF hot path.

The reference (DoubiZhizun/NeRF, package `nerf`) cannot run here (Scala + DJL + libtorch natives, no JVM,
no network) and ships no fixtures, so these vectors come from an INDEPENDENT op-by-op transliteration of
the Scala source onto torch CPU ops -- the same libtorch op families DJL's PyTorch engine dispatches
NDArray calls to.  Every function below follows the cited Scala lines in the reference's own
sample-major layout ([sample][ray][..]); fixtures are stored ray-major.  Gradients come from torch
autograd (DJL's GradientCollector is libtorch autograd).

This script is NOT part of the product or of the oracle; it exists so that the oracle
(oracle/nerf_oracle.cpp, a hand-written loop restatement) is checked against a second, independently
written implementation.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

torch.set_num_threads(4)


def position_code(x, L):  # nerfCoreBlock.scala:127-136
    out = [x]
    for i in range(L):
        y = x * (1 << i)
        out += [torch.sin(y), torch.cos(y)]
    return torch.cat(out, -1)


def sh2(viewdir):  # nerfCoreBlock.scala:138-174
    x, y, z = viewdir[..., 0], viewdir[..., 1], viewdir[..., 2]
    cosPhi = z
    sinPhi = torch.sqrt(-(z.square() - 1))
    ones, zeros = torch.ones_like(x), torch.zeros_like(y)
    cosTheta = torch.where(sinPhi < 1e-5, ones, x / sinPhi)
    sinTheta = torch.where(sinPhi < 1e-5, zeros, y / sinPhi)
    stcp, stsp = sinTheta * cosPhi, sinTheta * sinPhi
    out = [ones, cosTheta, stcp, stsp, cosTheta.square() - (1.0 / 3), stcp * cosTheta, stsp * cosTheta,
           (stcp * cosPhi * 2 - sinTheta) * sinTheta, stsp * sinTheta * cosPhi]
    return torch.stack(out, -1)


def core_block(c, p, pos, dirs, training, is_coarse):  # nerfCoreBlock.scala:17-112
    enc = position_code(pos, c["posL"])
    h = enc
    for i in range(c["D"]):
        h = torch.relu(F.linear(h, p[f"w{i}"], p[f"b{i}"].reshape(-1)))
        if i in c["skips"]:
            h = torch.cat([h, enc], -1)
    d = F.linear(h[:-1], p["ws"], p["bs"].reshape(-1))
    if is_coarse and not training:
        return d, None
    if c["useDir"] and c["useSH"]:
        q = F.linear(h, p["wf"], p["bf"].reshape(-1))
        q = q.reshape(q.shape[:-1] + (9, 3))
        rgb = (q * sh2(dirs).unsqueeze(-1)).sum(-2)
    elif c["useDir"]:
        a = F.linear(h, p["wf"], p["bf"].reshape(-1))
        b = F.linear(position_code(dirs, c["dirL"]), p["wd"], p["bd"].reshape(-1))
        rgb = F.linear(torch.relu(a + b), p["wc"], p["bc"].reshape(-1))
    else:
        rgb = F.linear(h, p["wf"], p["bf"].reshape(-1))
    return d, rgb


def get_weight(c, d, z, training, noise):  # nerfBlock.scala:31-45 (+ addNoise :21)
    dists = z[1:] - z[:-1]
    if training and c["rawNoiseStd"] > 0 and noise is not None:
        d = d + noise * c["rawNoiseStd"]
    alpha = (-torch.relu(d)) * dists
    T = torch.exp(torch.cumsum(alpha, 0))
    alpha = -(torch.exp(alpha) - 1)
    return torch.cat([alpha[:1], T[:-1] * alpha[1:], T[-1:]], 0)


def add_bkgd(c, rgb, w):  # nerfBlock.scala:24-29
    return rgb + (-(w.sum(0) - 1)) if c["whiteBkgd"] else rgb


def get_input(c, o, d, bounds, t_rand):  # nerfBlock.scala:47-69
    t = torch.linspace(0, 1, c["NSamples"]).unsqueeze(-1)
    near, far = bounds[..., 0], bounds[..., 1]
    if c["linDisp"]:
        z = 1 / ((1 / near) * (-(t - 1)) + (1 / far) * t)
    else:
        z = near * (-(t - 1)) + far * t
    if c["perturb"]:
        mids = (z[1:] + z[:-1]) * .5
        upper = torch.cat([mids, z[-1:]], 0)
        lower = torch.cat([z[:1], mids], 0)
        z = lower + (upper - lower) * t_rand
    z = z.unsqueeze(-1)
    return o + d * z, z


def sample_pdf(c, w, z, o, d, u_rand):  # nerfBlock.scala:71-106
    NI = c["NImportance"]
    wc = torch.cumsum(w.detach()[1:-1] + 1e-5, 0)
    s = wc[-1:]
    cdf = torch.cat([torch.zeros_like(s), wc / s], 0)
    bins = (z[1:] + z[:-1]) * .5
    B = cdf.shape[1]
    if c["perturb"]:
        u = torch.sort(u_rand, 0)[0]
    else:
        u = torch.linspace(0, 1, NI).repeat_interleave(B).reshape(NI, B, 1)
    cat = torch.cat([cdf, u], 0)
    inds = torch.argsort(torch.argsort(cat, dim=0, stable=True), dim=0, stable=True)[cdf.shape[0]:]
    inds = inds - torch.arange(NI).unsqueeze(-1).unsqueeze(-1)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[0] - 1)
    g0, g1 = torch.gather(cdf, 0, below), torch.gather(cdf, 0, above)
    b0, b1 = torch.gather(bins, 0, below), torch.gather(bins, 0, above)
    den = g1 - g0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - g0) / den
    zs = b0 + t * (b1 - b0)
    samples = torch.sort(torch.cat([zs, z], 0), 0)[0]
    return o + d * samples, samples, cdf, inds, zs


def forward(c, nets, o, d, bounds, viewdir, training, rand):  # nerfBlock.scala:108-127
    out = {}
    if c["useHierarchical"]:
        pc, pf = nets
        cpos, cz = get_input(c, o, d, bounds, rand.get("t_rand"))
        cd, crgb = core_block(c, pc, cpos, viewdir, training, True)
        cw = get_weight(c, cd, cz, training, rand.get("noise_c"))
        fpos, fz, cdf, inds, zs = sample_pdf(c, cw, cz, o, d, rand.get("u_rand"))
        fd, frgb = core_block(c, pf, fpos, viewdir, training, False)
        fw = get_weight(c, fd, fz, training, rand.get("noise_f"))
        fine = add_bkgd(c, (fw * torch.sigmoid(frgb)).sum(0), fw)
        coarse = add_bkgd(c, (cw * torch.sigmoid(crgb)).sum(0), cw) if training else None
        out.update(z_coarse=cz, w_coarse=cw, cdf=cdf, ind=inds, z_samples=zs, sigma_c=cd)
        if training:
            out.update(craw_c=crgb)
    else:
        (pf,) = nets
        fpos, fz = get_input(c, o, d, bounds, rand.get("t_rand"))
        fd, frgb = core_block(c, pf, fpos, viewdir, training, False)
        fw = get_weight(c, fd, fz, training, rand.get("noise_f"))
        fine = add_bkgd(c, (fw * torch.sigmoid(frgb)).sum(0), fw)
        coarse = None
        out.update(z_coarse=fz)
    out.update(rgb_fine=fine, rgb_coarse=coarse, z_fine=fz, w_fine=fw, sigma_f=fd, craw_f=frgb,
               depth=(fw * fz).sum(0).squeeze(-1), acc=fw.sum(0).squeeze(-1))  # nerf_origin/nerf_model.scala:217-219
    return out


def to_ray_major(t):
    if t is None:
        return None
    t = t.detach()
    if t.dim() == 3:  # [S][B][k] -> [B][S][k] (k==1 squeezed)
        t = t.permute(1, 0, 2)
        if t.shape[-1] == 1:
            t = t[..., 0]
    return t.contiguous().numpy()


def run_case(name, B, seed, training):
    c = H.CONFIGS[name]
    o, d, bounds, label = H.make_rays(B, seed)
    flat = H.make_params(c, seed)
    tflat = torch.from_numpy(flat.copy()).requires_grad_(training)
    nets = H.split_params(c, tflat)
    N, NI = c["NSamples"], c["NImportance"]
    S = N + NI if c["useHierarchical"] else N
    rand_np = {}
    if c["perturb"]:
        rand_np["t_rand"] = H.det_uniform(B * N, seed + 7).reshape(B, N)
        rand_np["u_rand"] = H.det_uniform(B * max(NI, 1), seed + 8).reshape(B, max(NI, 1))
    if training and c["rawNoiseStd"] > 0:
        rand_np["noise_c"] = H.det_normal(B * N, seed + 9).reshape(B, N)
        rand_np["noise_f"] = H.det_normal(B * S, seed + 10).reshape(B, S)
    rand = {}
    for k, v in rand_np.items():  # to sample-major [S][B][1]
        t = torch.from_numpy(v.T.copy()).unsqueeze(-1)
        if k.startswith("noise"):
            t = t[:-1]  # density exists for S-1 samples only
        elif k == "t_rand":
            t = t[..., 0]  # perturb acts before expandDims(-1) (nerfBlock.scala:64)
        rand[k] = t
    to = lambda a: torch.from_numpy(a)
    out = forward(c, nets, to(o), to(d), to(bounds), to(d), training, rand)
    res = {"rays_o": o, "rays_d": d, "bounds": bounds, "label": label, "B": np.int64(B), "seed": np.int64(seed),
           "training": np.int64(training)}
    for k, v in rand_np.items():
        res["rand_" + k] = v
    for k, v in out.items():
        a = to_ray_major(v)
        if a is not None:
            res[k] = a.astype(np.int32) if k == "ind" else a.astype(np.float32)
    if training:  # runNerf.scala:64,72-73
        lab = to(label)
        lf = ((lab - out["rgb_fine"]) ** 2).mean()
        loss = lf
        res["loss_fine"] = np.float32(lf.item())
        if c["useHierarchical"]:
            lc = ((lab - out["rgb_coarse"]) ** 2).mean()
            loss = loss + lc
            res["loss_coarse"] = np.float32(lc.item())
        loss.backward()
        g = tflat.grad.numpy().astype(np.float32)
        if g.size <= 200000:
            res["grads"] = g
        else:  # too large to commit: per-tensor norms and leading entries
            norms, heads = [], []
            off = 0
            for n in range(2 if c["useHierarchical"] else 1):
                for nm, r, k in H.layout(c):
                    blk = g[off:off + r * k]
                    norms.append(np.sqrt((blk.astype(np.float64) ** 2).sum()))
                    heads.append(np.resize(blk[:8], 8))
                    off += r * k
            res["grad_norms"] = np.array(norms, np.float64)
            res["grad_heads"] = np.stack(heads).astype(np.float32)
    return res


def misc_case():
    """Stand-alone op vectors: linspace, positionCode, SH2 (incl. the sinPhi<1e-5 guard), u8 rows."""
    res = {}
    for n in (2, 5, 64, 128, 192):
        res[f"linspace_{n}"] = torch.linspace(0, 1, n).numpy()
    x = torch.from_numpy(H.det_uniform(30, 5, -5, 5).reshape(10, 3))
    res["pc_x"] = x.numpy()
    res["pc_out_L10"] = position_code(x, 10).numpy()
    d = torch.from_numpy(H.det_uniform(30, 6, -1, 1).reshape(10, 3).astype(np.float32))
    d = d / d.norm(dim=-1, keepdim=True)
    d = torch.cat([d, torch.tensor([[0, 0, 1.0], [0, 0, -1.0], [1.0, 0, 0], [0, 1.0, 0]])], 0)
    res["sh_dir"] = d.numpy()
    res["sh_out"] = sh2(d).numpy()
    rgb = torch.from_numpy(H.det_uniform(60, 7, 0.0, 1.0).reshape(20, 3))
    res["u8_in"] = rgb.numpy()
    res["u8_out"] = (rgb * 255).to(torch.uint8).numpy()  # runNerf.scala:194
    return res


CASES = [
    ("small_dir", 8, 1, False), ("small_dir", 8, 1, True),
    ("small_sh", 8, 2, False), ("small_sh", 8, 2, True),
    ("small_nodir", 6, 3, True),
    ("small_flat", 8, 4, False), ("small_flat", 8, 4, True),
    ("small_opts", 8, 5, True), ("small_opts", 8, 5, False),
    ("full", 4, 6, False), ("full", 4, 6, True),
]

if __name__ == "__main__":
    for name, B, seed, training in CASES:
        res = run_case(name, B, seed, bool(training))
        fn = os.path.join(HERE, f"{name}_{'train' if training else 'eval'}.npz")
        np.savez_compressed(fn, **res)
        print(fn, os.path.getsize(fn))
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc_case())
