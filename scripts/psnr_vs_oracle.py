"""Full-size PSNR parity of the GPU training step against the CPU oracle (north star: "PSNR after a fixed number of steps
within 0.1 dB"; tests/test_gpu_train.py: test_psnr_after_fixed_steps is the same comparison on a small network and ONE batch).

Scene: a teacher NeRF (full 8x256 coarse+fine network, random weights) rendered in fp32 by this library from 4 training views
and one held-out view.  Students -- the full network of BASELINE configs[1], identical initial weights -- are trained for `steps`
Adam steps on the SAME sequence of `rays`-ray mini-batches by the oracle (oracle/nerf_oracle.cpp: hand backward + DJL-form Adam
on the CPU) and by the library in f32, f16x3 and bf16; at every quarter the held-out view is rendered by whoever trained the
weights and its PSNR printed.  usage: python scripts/psnr_vs_oracle.py [steps] [rays] [hw]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig
from oracle.oracle import Oracle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 40
nviews = 4
c = dict(H.CONFIGS["full"])
n = hw * hw
focal = 0.5 * hw / math.tan(0.5 * 0.6911112070083618)


def pose(th, ph=math.pi / 6, r=4.0):
    cam = r * np.array([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)])
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return np.stack([right, up, -fwd, cam], 1).astype(np.float32)


teacher = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n)))
teacher.set_params(H.make_params(c, 11, bias_scale=0.3))
f32 = dict(dtype=torch.float32, device="cuda")
views = []
for k in range(nviews + 1):
    ro, rd, bd = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 2, **f32)
    teacher.k_gen_rays(hw, hw, focal, pose(2 * math.pi * k / (nviews + 1) + 0.2), 2.0, 6.0, False, ro, rd, bd)
    img = torch.empty(n, 3, **f32)
    teacher.render_dev(ro, rd, bd, rd, n, img)
    teacher.synchronize()
    views.append(tuple(x.cpu().numpy() for x in (ro, rd, bd, img)))
teacher.close()
held = views[-1]
tr = [np.concatenate([v[i] for v in views[:-1]]) for i in range(4)]
print("teacher image: mean %.3f std %.3f; %d training rays, %d held-out rays" % (held[3].mean(), held[3].std(), tr[0].shape[0], n), flush=True)
init = H.make_params(c, 5, bias_scale=0.0)
rs = np.random.RandomState(7)
batches = [rs.randint(0, tr[0].shape[0], B) for _ in range(steps)]
marks = sorted({max(1, steps * q // 4) for q in (1, 2, 3, 4)})
psnr = lambda im: -10 * math.log10(float(((im - held[3]) ** 2).mean()))

curves, finals = {}, {}
o = Oracle(c)
p = init.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
t0 = time.perf_counter(); cur = []
for s, idx in enumerate(batches):
    r = o.train_grads(p, tr[0][idx], tr[1][idx], tr[2][idx], tr[1][idx], tr[3][idx])
    o.adam(p, r["grads"], m, v, s + 1)
    if s + 1 in marks:
        cur.append((s + 1, psnr(o.forward(p, held[0], held[1], held[2], held[1])["rgb_fine"]), r["loss_fine"]))
        print("  oracle step %d: %.3f dB (%.0f s)" % (s + 1, cur[-1][1], time.perf_counter() - t0), flush=True)
curves["oracle"] = cur; finals["oracle"] = p.copy()
for dtype in ("f32", "f16x3", "bf16"):
    st = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=max(B, n), dtype=dtype)))
    st.set_params(init)
    cur = []
    for s, idx in enumerate(batches):
        lf, lc = st.train_step(tr[0][idx], tr[1][idx], tr[2][idx], tr[1][idx], tr[3][idx])
        if s + 1 in marks:
            cur.append((s + 1, psnr(st.evaluate(held[0], held[1], held[2], held[1])), lf))
    curves[dtype] = cur; finals[dtype] = st.get_params()
    st.close()
print("held-out PSNR (dB) at steps", [x[0] for x in curves["oracle"]])
for k, cur in curves.items():
    print("  %-6s" % k, " ".join("%.3f" % x[1] for x in cur), "  last batch loss_fine %.6f" % cur[-1][2])
for k in ("f32", "f16x3", "bf16"):
    d = [abs(a[1] - b[1]) for a, b in zip(curves[k], curves["oracle"])]
    print("  %-6s max |dPSNR vs oracle| %.4f dB; parameters after %d steps: max |dp| %.2e (|p| max %.2f)" %
          (k, max(d), steps, float(np.abs(finals[k] - finals["oracle"]).max()), float(np.abs(finals["oracle"]).max())))
