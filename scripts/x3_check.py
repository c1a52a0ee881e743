"""split-f16 render kernel vs the fp32 kernel and the oracle: MLP stage error and render timing (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import gpu_util as G
from oracle.oracle import Oracle
names = sys.argv[1:] or ["full", "small_dir", "small_sh", "small_nodir"]
for name in names:
    c = H.CONFIGS[name]
    if c["W"] == 32: continue
    n = 300
    ro, rd, b, _ = H.make_rays(n, 3)
    params = H.make_params(c, 2)
    S = c["NSamples"] + (c["NImportance"] if c["useHierarchical"] else 0)
    z = np.sort(H.det_uniform(n * S, 5, 2.0, 6.0).reshape(n, S), axis=1)
    net = 1 if c["useHierarchical"] else 0
    o = Oracle(c); npc = o.net_param_count()
    rs, rc = o.mlp_forward(params[npc * net:npc * (net + 1)], False, True, ro, rd, rd, z)
    for dt in ("f32", "f16x3"):
        blk = G.make_block(c, dtype=dt); blk.set_params(params)
        sig, craw = G.empty(n, S), G.empty(n, S, 3)
        blk.k_mlp_forward(net, G.dev(ro), G.dev(rd), G.dev(rd), G.dev(z), n, S, sig, craw)
        blk.synchronize()
        es = np.abs(sig.cpu().numpy()[:, :S-1] - rs[:, :S-1]); ec = np.abs(craw.cpu().numpy() - rc)
        tol_s = (es / (2e-4 * np.abs(rs[:, :S-1]) + 2e-5)).max(); tol_c = (ec / (2e-4 * np.abs(rc) + 2e-5)).max()
        print(f"{name:12s} {dt:6s} sigma max err {es.max():.2e} (x tol {tol_s:.3f})  craw max err {ec.max():.2e} (x tol {tol_c:.3f})", flush=True)
        fine = blk.evaluate(ro, rd, b, rd)
        ref = o.forward(params, ro, rd, b, rd)["rgb_fine"]
        print(f"{'':12s} {dt:6s} render max |rgb - oracle| {np.abs(fine - ref).max():.2e}", flush=True)
        blk.close()
if "full" in names:
    c = H.CONFIGS["full"]; n = 160000
    ro, rd, b, _ = H.make_rays(n, 3)
    dro, drd, db = G.dev(ro), G.dev(rd), G.dev(b); out = G.empty(n, 3)
    for dt in ("f32", "bf16", "f16x3"):
        blk = G.make_block(c, max_rays=n, dtype=dt); blk.init_params(1)
        blk.render_dev(dro, drd, db, drd, n, out); blk.synchronize()
        blk.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(2): blk.render_dev(dro, drd, db, drd, n, out)
        blk.synchronize(); dt_ = (time.perf_counter() - t0) / 2
        p = blk.get_profile()
        print(f"render 160000 rays {dt}: {dt_*1e3:.1f} ms  {n/dt_/1e3:.0f} k rays/s  mlp fwd {p['mlp_fwd_ms']/2:.1f} ms = {p['mlp_fwd_flop']/p['mlp_fwd_ms']/1e9:.0f} TF", flush=True)
        blk.close()
