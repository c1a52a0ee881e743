"""one-off: the bf16 TRAINING kernels (forward, backward-data, weight gradient) against the torch emulation of their arithmetic
(tests/bf16_ref.py, evaluated at the device's fine depths) over the random configurations of tests/test_gpu_fuzz.py's
generator with W >= 64; bars of tests/test_gpu_bf16.py: loss 1e-2 relative, whole gradient 2e-2, every tensor 1e-1 of
max(its norm, 5 % of the largest tensor's).
usage: fuzz_bf16_grads.py <first seed> <last seed + 1> [device]     device: the emulation encodes the sample positions with the
kernels' argument rounding (bf16_ref.position_code_device) instead of the exact sine"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, gpu_util as G
import test_gpu_fuzz as F
import bf16_ref

DEV = len(sys.argv) > 3 and sys.argv[3] == "device"
bad = cases = 0
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    c, n, r = F.draw(seed)
    if c["W"] < 64:
        continue
    c = dict(c, perturb=False, rawNoiseStd=0.0)
    ro, rd, b, label = H.make_rays(n, 200 + seed)
    params = H.make_params(c, 50 + seed)
    blk = G.make_block(c, max_rays=max(n, 8), dtype="bf16")
    blk.set_params(params)
    lf, lc, dbg = blk.train_backward(ro, rd, b, rd, label, debug=True)
    gr = blk.get_grads()
    blk.close()
    rlf, rlc, rg = bf16_ref.train_grads(c, params, ro, rd, b, label, None, z_fine_in=dbg["z_fine"] if c["useHierarchical"] else None,
                                          device_encoding=DEV)
    whole = float(np.linalg.norm(gr - rg) / np.linalg.norm(rg))
    spans = H.param_spans(c)
    big = max(np.linalg.norm(rg[o:o + k]) for _, _, o, k in spans)
    tens = max(float(np.linalg.norm(gr[o:o + k] - rg[o:o + k]) / max(np.linalg.norm(rg[o:o + k]), 0.05 * big)) for _, _, o, k in spans)
    dl = abs(lf - rlf) / abs(rlf)
    if c["useHierarchical"]:
        dl = max(dl, abs(lc - rlc) / abs(rlc))
    ok = np.isfinite(gr).all() and whole < 2e-2 and tens <= 1e-1 and dl <= 1e-2
    cases += 1; bad += not ok; worst = max(worst, whole)
    print(seed, "ok" if ok else "OVER", "whole %.2e worst tensor %.2e loss %.2e" % (whole, tens, dl),
          {k: c[k] for k in ("D", "W", "skips", "posL", "dirL", "useDir", "useSH", "useHierarchical", "NSamples", "NImportance")}, n, flush=True)
print("cases", cases, "over the bars", bad, "worst whole", worst)
