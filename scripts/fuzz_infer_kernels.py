"""one-off: tests/test_gpu_infer_kernels.py's comparison (the 16x16x32 inference kernel against the 32x32x16 training forward at
the SAME depths, same parameters) over the random configurations of tests/test_gpu_fuzz.py's generator, f16x3 and bf16.
usage: fuzz_infer_kernels.py <first seed> <last seed + 1>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, gpu_util as G
import test_gpu_fuzz as F
import test_gpu_infer_kernels as T

worst = {"f16x3": 0.0, "bf16": 0.0, "bf16_mean": 0.0}
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    c, n, r = F.draw(seed)
    if c["W"] < 64:
        continue
    c = dict(c, rawNoiseStd=0.0, perturb=False)
    ro, rd, b, label = H.make_rays(n, 200 + seed)
    params = H.make_params(c, 50 + seed)
    line = []
    for dtype in ("f16x3", "bf16"):
        blk = G.make_block(c, max_rays=max(n, 8), dtype=dtype)
        blk.set_params(params)
        sig_t, craw_t, sig_i, craw_i = T._stage(blk, c, ro, rd, b, label)
        blk.close()
        S = sig_t.shape[1]
        dmax = dmean = 0.0
        for a, ref in ((sig_i[:, :S - 1], sig_t[:, :S - 1]), (craw_i, craw_t)):
            d = np.abs(a - ref) / max(np.abs(ref).max(), 1e-30)
            dmax = max(dmax, float(d.max())); dmean = max(dmean, float(d.mean()))
        ok = dmax <= 1e-5 if dtype == "f16x3" else (dmax <= 2e-2 and dmean <= 1e-4)
        bad += not ok
        worst[dtype] = max(worst[dtype], dmax)
        if dtype == "bf16":
            worst["bf16_mean"] = max(worst["bf16_mean"], dmean)
        line.append("%s %s max %.2e mean %.2e" % (dtype, "ok" if ok else "OVER", dmax, dmean))
    print(seed, " | ".join(line), {k: c[k] for k in ("D", "W", "skips", "posL", "dirL", "useDir", "useSH", "useHierarchical", "NSamples", "NImportance")}, n, flush=True)
print("worst", worst, "over the bars:", bad)
