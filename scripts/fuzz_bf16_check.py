"""one-off: where a LARGE bf16 render difference against the fp32 oracle comes from.  For seeds of tests/test_gpu_fuzz.py's
generator: the device's bf16 eval render against (a) the fp32 oracle and (b) the bf16 EMULATION of the same arithmetic
(tests/bf16_ref.py: every Linear's operands rounded to bf16, fp32 accumulation) -- if (b) is small and (a) is large, the
difference is what bf16 operands do to that (random, untrained, coarsely sampled) network, not a kernel defect.
    python scripts/fuzz_bf16_check.py <seed> [<seed> ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H, gpu_util as G
from oracle.oracle import Oracle
import test_gpu_fuzz as F
import bf16_ref as R

to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
for seed in map(int, sys.argv[1:]):
    c, n, r = F.draw(seed)
    c = dict(c, perturb=False)
    ro, rd, b, label = H.make_rays(n, 200 + seed)
    params = H.make_params(c, 50 + seed)
    blk = G.make_block(c, max_rays=max(n, 8), dtype="bf16")
    blk.set_params(params)
    fine, _, dbg = blk.forward(ro, rd, b, rd, training=False, rand={}, debug=True)
    blk.close()
    ref = Oracle(c).forward(params, ro, rd, b, rd)
    nets = H.split_params(c, to(params))
    saved = R.mg.core_block
    R.mg.core_block = R.core_block_bf16
    try:
        with torch.no_grad():
            emu = R.mg.forward(c, nets, to(ro), to(rd), to(b), to(rd), False, {})
    finally:
        R.mg.core_block = saved
    e_rgb = emu["rgb_fine"].numpy()
    ea = np.abs(fine - ref["rgb_fine"]); eb = np.abs(fine - e_rgb); ec = np.abs(e_rgb - ref["rgb_fine"])
    zf_e = R.mg.to_ray_major(emu["z_fine"])
    same_z = float((np.asarray(zf_e).reshape(dbg["z_fine"].shape) == dbg["z_fine"]).mean())
    print(f"seed {seed}: device bf16 vs fp32 oracle max {ea.max():.3e} (ray {int(ea.max(1).argmax())}, mean {ea.mean():.2e});  "
          f"device bf16 vs bf16 emulation max {eb.max():.3e} (mean {eb.mean():.2e});  emulation vs fp32 oracle max {ec.max():.3e};  "
          f"fine depths equal to the emulation's: {100 * same_z:.1f} %   "
          f"{ {k: c[k] for k in ('D', 'W', 'skips', 'posL', 'dirL', 'useDir', 'useSH', 'useHierarchical', 'NSamples', 'NImportance')} } n={n}",
          flush=True)
