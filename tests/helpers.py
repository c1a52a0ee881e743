"""Shared, dependency-free test helpers: deterministic parameters / rays and named configurations.

Everything here is reproducible integer arithmetic (splitmix64) so fixtures can be regenerated
bit-identically on any machine without relying on a library RNG's stream stability.
"""
import math

import numpy as np

BASE_CFG = dict(useDir=True, useSH=False, useHierarchical=True, posL=10, dirL=4, D=8, W=256, skips=[4],
                NSamples=64, NImportance=128, rawNoiseStd=0.0, whiteBkgd=True, linDisp=False, perturb=False,
                lrate=5e-4, lrateDecay=250)


def cfg(**kw):
    c = dict(BASE_CFG)
    c.update(kw)
    return c


# named configurations used by golden fixtures and parity tests
CONFIGS = {
    # small networks (fast on the CPU oracle), one per colour head / option
    "small_dir": cfg(D=4, W=64, skips=[1], posL=4, dirL=2, NSamples=16, NImportance=24),
    "small_sh": cfg(D=4, W=64, skips=[1], posL=4, dirL=2, NSamples=16, NImportance=24, useSH=True),
    "small_nodir": cfg(D=3, W=32, skips=[], posL=3, dirL=2, NSamples=16, NImportance=8, useDir=False,
                       whiteBkgd=False),
    "small_flat": cfg(D=4, W=64, skips=[2], posL=4, dirL=2, NSamples=32, NImportance=0, useSH=True,
                      useHierarchical=False),
    "small_opts": cfg(D=4, W=64, skips=[1], posL=4, dirL=2, NSamples=16, NImportance=24, linDisp=True,
                      perturb=True, rawNoiseStd=1.0, whiteBkgd=False),
    # the full network of BASELINE.json configs[1]
    "full": cfg(),
    # the reference's shipped default (runNerf.scala:24-56): SH head, no hierarchical pass
    "shipped": cfg(useSH=True, useHierarchical=False, NImportance=64, rawNoiseStd=1.0),
}


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def det_uniform(n, seed, lo=0.0, hi=1.0):
    """n float32 values in [lo, hi): hash of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        h = splitmix64(splitmix64(idx))
    u = ((h >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32)


def det_normal(n, seed):
    u1 = det_uniform(n, seed * 2 + 1).astype(np.float64)
    u2 = det_uniform(n, seed * 2 + 2).astype(np.float64)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * math.pi * u2)).astype(np.float32)


def is_skip(c, i):
    return i in c["skips"]


def layout(c):
    """Canonical flat parameter layout of ONE network (DJL child order): list of (name, rows, cols)."""
    E = 3 * (1 + 2 * c["posL"])
    Ed = 3 * (1 + 2 * c["dirL"])
    W, D = c["W"], c["D"]
    items = []
    for l in range(D):
        fan = E if l == 0 else W + (E if is_skip(c, l - 1) else 0)
        items += [(f"w{l}", W, fan), (f"b{l}", 1, W)]
    fw = W + (E if is_skip(c, D - 1) else 0)
    items += [("ws", 1, fw), ("bs", 1, 1)]
    if c["useDir"] and not c["useSH"]:
        items += [("wf", W // 2, fw), ("bf", 1, W // 2), ("wd", W // 2, Ed), ("bd", 1, W // 2),
                  ("wc", 3, W // 2), ("bc", 1, 3)]
    elif c["useDir"]:
        items += [("wf", 27, fw), ("bf", 1, 27)]
    else:
        items += [("wf", 3, fw), ("bf", 1, 3)]
    return items


def net_param_count(c):
    return sum(r * k for _, r, k in layout(c))


def param_count(c):
    return net_param_count(c) * (2 if c["useHierarchical"] else 1)


def make_params(c, seed=0, bias_scale=0.1):
    """He-uniform weights (variance 2/fan_in, as DJL's default initialiser) and small non-zero biases
    (so bias paths are exercised), flat canonical order: coarse network first when hierarchical."""
    out = []
    nets = 2 if c["useHierarchical"] else 1
    s = seed * 1000 + 17
    for n in range(nets):
        for name, rows, cols in layout(c):
            s += 1
            if name.startswith("w"):
                lim = math.sqrt(6.0 / cols)
                out.append(det_uniform(rows * cols, s, -lim, lim))
            else:
                out.append(det_uniform(rows * cols, s, -bias_scale, bias_scale))
    return np.concatenate(out).astype(np.float32)


def split_params(c, flat):
    """-> list (per network) of dict name -> 2-D array view."""
    nets = 2 if c["useHierarchical"] else 1
    res = []
    off = 0
    for n in range(nets):
        d = {}
        for name, rows, cols in layout(c):
            d[name] = flat[off:off + rows * cols].reshape(rows, cols)
            off += rows * cols
        res.append(d)
    assert off == flat.shape[0]
    return res


def make_rays(B, seed=0, radius=4.0, near=2.0, far=6.0):
    """Rays from cameras on a sphere of `radius` looking roughly at the origin (Blender-style scene,
    getDataSet.scala:62-63 near=2 far=6).  Returns rays_o, unit rays_d, bounds, and a label image."""
    u = det_uniform(B * 8, seed + 101).reshape(B, 8).astype(np.float64)
    theta = 2 * math.pi * u[:, 0]
    phi = np.arccos(1 - 1.2 * u[:, 1])          # upper hemisphere-ish
    o = radius * np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)], -1)
    target = (u[:, 2:5] - 0.5) * 1.6            # aim somewhere inside the unit-ish cube
    d = target - o
    n = np.linalg.norm(d, axis=-1, keepdims=True)
    d = d / n
    # emulate "bounds = norm * (near, far)" with norm of an un-normalised pixel direction in [1, 1.2]
    nrm = 1.0 + 0.2 * u[:, 5:6]
    bounds = np.concatenate([nrm * near, nrm * far], -1)
    label = u[:, 5:8]
    f32 = np.float32
    return o.astype(f32), d.astype(f32), bounds.astype(f32), label.astype(f32)


def param_spans(c):
    """[(network, tensor name, offset, count)] of the flat parameter / gradient vector"""
    out, off = [], 0
    for net in range(2 if c["useHierarchical"] else 1):
        for nm, r, k in layout(c):
            out.append((net, nm, off, r * k))
            off += r * k
    return out


def assert_grads_close(c, gr, rg, whole=1e-3, per_tensor=5e-3, floor=0.02):
    """Whole gradient vector within `whole` (relative L2) and EVERY tensor within `per_tensor` of its own norm -- or of
    `floor` x the largest tensor's norm for the tiny ones (a bias, the density row), so that a wrong small tensor cannot
    hide behind the large ones the way a norm-vs-norm comparison lets it."""
    assert np.isfinite(gr).all()
    assert np.linalg.norm(gr - rg) / np.linalg.norm(rg) < whole, np.linalg.norm(gr - rg) / np.linalg.norm(rg)
    spans = param_spans(c)
    big = max(np.linalg.norm(rg[o:o + k]) for _, _, o, k in spans)
    for net, nm, o, k in spans:
        err = np.linalg.norm(gr[o:o + k] - rg[o:o + k])
        assert err <= per_tensor * max(np.linalg.norm(rg[o:o + k]), floor * big), (net, nm, err, np.linalg.norm(rg[o:o + k]))
