"""one-off: the render entry points against each other over the random configurations of tests/test_gpu_fuzz.py's generator,
every dtype the width allows.  A ray's colour is a function of that ray alone, so all of these must agree BIT FOR BIT:
  (1) nerfhip_render of 3n rays on a handle whose max_rays is a random number below 3n (blocks of max_rays, the last one
      ragged) against a handle that takes them in one block;
  (2) nerfhip_render_u8 on both;
  (3) nerfhip_render_dev_static (no allocation, static scratch) against nerfhip_render_dev, twice, the second call smaller;
  (4) nerfhip_render_pose_u8 of a random small frame against nerfhip_k_gen_rays + nerfhip_render_u8, and against two
      nerfhip_render_pose_rows_u8 calls split at a random row (NDC or not).
usage: fuzz_entry_points.py <first seed> <last seed + 1>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H, gpu_util as G
import test_gpu_fuzz as F

bad = 0
cases = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    c, n, r = F.draw(seed)
    c = dict(c, perturb=False)
    n3 = 3 * n
    ro, rd, b, _ = H.make_rays(n3, 300 + seed)
    params = H.make_params(c, 60 + seed)
    mr = int(r.randint(1, n3 + 1))
    hh, ww = int(r.randint(2, 13)), int(r.randint(2, 21))
    split = int(r.randint(0, hh + 1))
    ndc = bool(r.randint(0, 2)) and not c["linDisp"]   # 1 / near with the NDC near plane of 0 is infinite in the reference too
    # a camera near (0, 0, 4) looking down -z at the origin, turned by up to 0.6 / 0.3 rad about y / x (the scenes' geometry:
    # an arbitrary rotation sends NDC rays through d_z ~ 0 and the positions out of every arithmetic's range)
    ty, tx = r.uniform(-0.6, 0.6), r.uniform(-0.3, 0.3)
    Ry = np.array([[np.cos(ty), 0, np.sin(ty)], [0, 1, 0], [-np.sin(ty), 0, np.cos(ty)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(tx), -np.sin(tx)], [0, np.sin(tx), np.cos(tx)]])
    c2w = np.concatenate([Ry @ Rx, r.uniform(-0.5, 0.5, (3, 1)) + np.array([[0], [0], [4.0]])], 1).astype(np.float32)
    focal = float(r.uniform(0.8, 2.0) * ww)
    near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
    msgs = []
    for dtype in (("f32", "f16x3", "bf16") if c["W"] >= 64 else ("f32",)):
        big = G.make_block(c, max_rays=max(n3, hh * ww), dtype=dtype)
        small = G.make_block(c, max_rays=mr, dtype=dtype)
        try:
            for x in (big, small):
                x.set_params(params)
            ref = big.evaluate(ro, rd, b, rd)
            ok1 = np.array_equal(small.evaluate(ro, rd, b, rd), ref)
            ref8 = big.render_u8(ro, rd, b, rd)
            ok2 = np.array_equal(small.render_u8(ro, rd, b, rd), ref8)
            ok3 = True
            if n3 <= small.render_static_max_rays():
                dro, drd, db = G.dev(ro), G.dev(rd), G.dev(b)
                for k in (n3, n):
                    got = torch.full((k, 3), -7.0, device="cuda")
                    small.render_dev_static(dro[:k], drd[:k], db[:k], drd[:k], k, got)
                    small.synchronize()
                    ok3 = ok3 and np.array_equal(got.cpu().numpy(), ref[:k])
            frame = big.render_pose(hh, ww, focal, c2w, near, far, ndc)
            go, gd, gb = G.empty(hh * ww, 3), G.empty(hh * ww, 3), G.empty(hh * ww, 2)
            big.k_gen_rays(hh, ww, focal, c2w, near, far, int(ndc), go, gd, gb)
            big.synchronize()
            via_rays = big.render_u8(go.cpu().numpy(), gd.cpu().numpy(), gb.cpu().numpy(), gd.cpu().numpy()).reshape(hh, ww, 3)
            ok4 = np.array_equal(frame, via_rays)
            rows = np.concatenate([small.render_pose_rows(hh, ww, focal, c2w, near, far, ndc, 0, split),
                                   small.render_pose_rows(hh, ww, focal, c2w, near, far, ndc, split, hh)], 0)
            ok5 = np.array_equal(rows, frame)
            oks = (ok1, ok2, ok3, ok4, ok5)
            cases += 1
            bad += not all(oks)
            msgs.append("%s %s" % (dtype, "ok" if all(oks) else "MISMATCH " + str(oks)))
        finally:
            big.close(); small.close()
    print(seed, " | ".join(msgs), {k: c[k] for k in ("D", "W", "skips", "posL", "dirL", "useDir", "useSH", "useHierarchical", "NSamples", "NImportance")},
          "rays", n3, "max_rays", mr, "frame", (hh, ww), "split", split, "ndc", ndc, flush=True)
print("cases", cases, "mismatches", bad)
