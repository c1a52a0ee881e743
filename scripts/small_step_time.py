"""Wall-clock against kernel time of a training step at the sizes the REFERENCE ships (runNerf.scala:24-56: 1024 rays,
64 samples, SH head, no hierarchical pass) and at the full 4096 x (64+128) batch: tells whether the host's launch stream
or the GPU bounds the step.    python scripts/small_step_time.py [steps] [library]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as H
from nerf_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from nerf_amd import NerfBlock, NerfConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = lambda a: torch.from_numpy(a).cuda()
out = {}
for name, n in (("shipped", 1024), ("shipped", 4096), ("full", 1024), ("full", 4096)):
    for dtype in ("f32", "f16x3", "bf16"):
        tr = NerfBlock(NerfConfig.from_dict(dict(H.CONFIGS[name], dtype=dtype, maxRays=n)))
        tr.init_params(1)
        ro, rd, b, lab = H.make_rays(n, 1)
        o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
        for _ in range(5):
            tr.train_step_dev(o, d, bb, d, l, n)
        tr.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step_dev(o, d, bb, d, l, n)
        t_issue = time.perf_counter() - t0      # the host is done issuing
        tr.synchronize()
        t_all = time.perf_counter() - t0
        # one step at a time, synchronised: what a caller that reads the loss every step (runNerf.scala:102) sees
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step_dev(o, d, bb, d, l, n)
            tr.synchronize()
        t_sync = time.perf_counter() - t0
        key = f"{name}_{n}_{dtype}"
        out[key] = {"ms_per_step": round(1e3 * t_all / steps, 4), "host_issue_ms": round(1e3 * t_issue / steps, 4),
                    "ms_per_step_synchronised": round(1e3 * t_sync / steps, 4),
                    "rays_per_s": round(n * steps / t_all, 1)}
        print(key, out[key], flush=True)
        del tr
print(json.dumps(out))
