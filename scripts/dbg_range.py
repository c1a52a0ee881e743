"""debug: what does the f16x3 forward produce when activations pass 65504?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import helpers as H, gpu_util as G
c = H.CONFIGS["small_dir"]
ro, rd, b, label = H.make_rays(32, 3)
p = H.make_params(c, 1)
for net, nm, off, k in H.param_spans(c):
    if nm in ("w0", "w1", "w2", "w3"):
        p[off:off + k] *= 60.0
for dt in ("f32", "f16x3"):
    blk = G.make_block(c, dtype=dt)
    blk.set_params(p)
    try:
        fine, coarse, dbg = blk.forward(ro, rd, b, rd, debug=True)
        print(dt, "rgb", fine[:2], "sigma_c", dbg["sigma_c"][0, :4], "sigma_f", dbg["sigma_f"][0, :4], "craw", dbg["craw_f"][0, :2],
              "finite:", np.isfinite(dbg["sigma_f"][:, :-1]).all(), np.isfinite(dbg["craw_f"]).all(), "max|sigma_f|", np.nanmax(np.abs(dbg["sigma_f"][:, :-1])))
    except Exception as e:
        print(dt, "raised", e)
    blk.close()
