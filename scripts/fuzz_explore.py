"""one-off: more seeds of tests/test_gpu_fuzz.py's generator, plus the bf16 render error of every configuration"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, gpu_util as G
from oracle.oracle import Oracle
import test_gpu_fuzz as F
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        F.test_random_configuration(seed)
        st = "ok"
    except AssertionError as e:
        st = "FAIL " + str(e)[:300]
    c, n, r = F.draw(seed)
    err = -1.0
    if c["W"] >= 64:
        ro, rd, b, label = H.make_rays(n, 200 + seed)
        params = H.make_params(c, 50 + seed)
        blk = G.make_block(dict(c, perturb=False), max_rays=max(n, 8), dtype="bf16")
        blk.set_params(params)
        fine = blk.evaluate(ro, rd, b, rd)
        ref = Oracle(dict(c, perturb=False)).forward(params, ro, rd, b, rd)["rgb_fine"]
        err = float(np.abs(fine - ref).max())
        blk.close()
        worst = max(worst, err)
    print(seed, st, "bf16 render err %.2e" % err, {k: c[k] for k in ("D", "W", "skips", "posL", "dirL", "useDir", "useSH", "useHierarchical", "NSamples", "NImportance")}, n, flush=True)
print("worst bf16", worst)
