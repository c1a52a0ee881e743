"""Emulation of the bf16-MFMA MLP arithmetic on top of the golden torch transliteration (tests/golden/make_golden.py):
every Linear of nerfCoreBlock rounds its input activations and its weights to bf16 (RNE) and accumulates in fp32;
its backward rounds the incoming gradient to bf16 and uses the same rounded operands -- exactly what
mlp_fwd_bf16 / mlp_bwd_bf16 / wgrad_bf16 do.  The per-ray direction branch runs in fp32 forward (dir_branch_kernel)
but its weight gradient uses bf16 operands.  Test infrastructure only."""
import importlib.util
import os

import numpy as np
import torch
import torch.nn.functional as F

import helpers as H

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def rnd(x):
    return x.to(torch.bfloat16).to(torch.float32)


class LinBf16(torch.autograd.Function):
    """y = rnd(x) rnd(w)^T + b;  backward with rnd(dy).  fwd_round=False: fp32 forward, bf16 backward operands."""

    @staticmethod
    def forward(ctx, x, w, b, fwd_round):
        xr, wr = rnd(x), rnd(w)
        ctx.save_for_backward(xr, wr)
        if fwd_round:
            return F.linear(xr, wr) + b.reshape(-1)
        return F.linear(x, w) + b.reshape(-1)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        g = rnd(gy)
        gx = g @ wr
        g2, x2 = g.reshape(-1, g.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return gx, g2.t() @ x2, g2.sum(0).reshape(1, -1), None


def position_code_device(x, L):
    """positionCode as the bf16 kernels evaluate it (csrc/bf16_common.hip.h: code_value_fast): the argument in revolutions,
    x 2^f fl32(1 / 2 pi) rounded to fp32 -- 6e-8 relative of up to 330 revolutions at f = 9, i.e. up to 1.2e-4 rad -- reduced by
    fract and handed to the hardware sine (taken as exact here).  An encoding value within that distance of a bf16 rounding
    boundary lands on the other bf16 value than the exact sine's does."""
    out = [x]
    inv2pi = np.float32(0.15915494309189535)
    for i in range(L):
        rev = (x * float(np.float32(1 << i) * inv2pi)).to(torch.float32)
        rev = (rev - torch.floor(rev)).to(torch.float64)
        out += [torch.sin(2 * np.pi * rev).to(torch.float32), torch.cos(2 * np.pi * rev).to(torch.float32)]
    return torch.cat(out, -1)


def lin(x, w, b, fwd_round=True):
    return LinBf16.apply(x, w, b, fwd_round)


_DEVICE_ENCODING = False


def core_block_bf16(c, p, pos, dirs, training, is_coarse):
    enc = (position_code_device if _DEVICE_ENCODING else mg.position_code)(pos, c["posL"])
    h = enc
    for i in range(c["D"]):
        h = torch.relu(lin(h, p[f"w{i}"], p[f"b{i}"]))
        if i in c["skips"]:
            h = torch.cat([h, enc], -1)
    d = lin(h[:-1], p["ws"], p["bs"])
    if is_coarse and not training:
        return d, None
    if c["useDir"] and c["useSH"]:
        q = lin(h, p["wf"], p["bf"])
        q = q.reshape(q.shape[:-1] + (9, 3))
        rgb = (q * mg.sh2(dirs).unsqueeze(-1)).sum(-2)
    elif c["useDir"]:
        a = lin(h, p["wf"], torch.zeros_like(p["bf"]))
        # the kernel adds (Wd enc(dir) + bd + bf) in fp32 per ray; gradients of wd/bd/bf from the bf16 dh
        b = lin(mg.position_code(dirs, c["dirL"]), p["wd"], p["bd"], False) + (p["bf"].reshape(-1) - p["bf"].reshape(-1).detach()) \
            + p["bf"].reshape(-1).detach()
        hc = torch.relu(a + b)
        rgb = lin(hc, p["wc"], p["bc"])
    else:
        rgb = lin(h, p["wf"], p["bf"])
    return d, rgb


def train_grads(c, flat, o, d, bounds, label, rand_np=None, z_fine_in=None, device_encoding=False):
    """-> (loss_fine, loss_coarse or None, flat gradient) of the bf16-emulated training forward/backward.
    z_fine_in ([rays][N + NImportance], e.g. the device's): the fine pass at THESE depths instead of the emulation's own
    resampling (the sampler carries no gradient, nerfBlock.scala:81; a coarse weight that differs in its last bits can move a
    fine sample across a bin, which is the sampler's discontinuity, not the arithmetic under test).
    device_encoding: the positional encoding of the SAMPLE POSITIONS with the kernels' argument rounding (position_code_device)."""
    rand_np = rand_np or {}
    tflat = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32).copy()).requires_grad_(True)
    nets = H.split_params(c, tflat)
    rand = {}
    for k, v in rand_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v.T)).unsqueeze(-1)
        if k.startswith("noise"):
            t = t[:-1]
        elif k == "t_rand":
            t = t[..., 0]
        rand[k] = t
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    global _DEVICE_ENCODING
    saved, saved_pdf = mg.core_block, mg.sample_pdf
    mg.core_block = core_block_bf16
    _DEVICE_ENCODING = bool(device_encoding)
    if z_fine_in is not None and c["useHierarchical"]:
        zf = torch.from_numpy(np.ascontiguousarray(np.asarray(z_fine_in, np.float32).T)).unsqueeze(-1)   # [sample][ray][1]
        mg.sample_pdf = lambda c_, w, z, o_, d_, u: (o_ + d_ * zf, zf, None, None, None)
    try:
        out = mg.forward(c, nets, to(o), to(d), to(bounds), to(d), True, rand)
    finally:
        mg.core_block, mg.sample_pdf = saved, saved_pdf
        _DEVICE_ENCODING = False
    lab = to(label)
    lf = ((lab - out["rgb_fine"]) ** 2).mean()
    loss, lc = lf, None
    if c["useHierarchical"]:
        lc = ((lab - out["rgb_coarse"]) ** 2).mean()
        loss = loss + lc
    loss.backward()
    return float(lf.detach()), (None if lc is None else float(lc.detach())), tflat.grad.numpy().astype(np.float32)
