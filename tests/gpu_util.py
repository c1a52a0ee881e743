"""Helpers for the -m gpu parity tests: device buffers via torch (plumbing only), oracle access."""
import numpy as np
import torch

import helpers as H
from nerf_amd import NerfBlock, NerfConfig


def dev(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def empty(*shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device="cuda")


def make_block(cfg_dict, max_rays=4096, **kw):
    c = NerfConfig.from_dict(cfg_dict)
    c.maxRays = max_rays
    for k, v in kw.items():
        setattr(c, k, v)
    return NerfBlock(c)


def close_frac(a, b, rtol, atol):
    return float((np.abs(a - b) <= atol + rtol * np.abs(b)).mean())
