"""Full-size PSNR comparison of fp32 and bf16 training (north star: "PSNR after a fixed number of steps within 0.1 dB").

Scene: a TEACHER NeRF (full 8x256 coarse+fine network, random N(0, 2/fan_in) weights scaled up so that the volume is
not empty) rendered in fp32 by this library from `nviews` poses on the radius-4 sphere = a view-consistent synthetic
scene of BASELINE's shape without any dataset.  STUDENTS (same initial weights, different from the teacher) are
trained for `steps` Adam steps of 4096 rays drawn from the training views with dtype f32, f16x3 and bf16, each twice;
PSNR is measured on a held-out view.  usage: python scripts/psnr_full.py [steps] [hw]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import NerfBlock, NerfConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nviews, B = 8, 4096
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
rng = np.random.default_rng(0)
tp = H.make_params(c, 11, bias_scale=0.3)
teacher.set_params(tp)
f32 = dict(dtype=torch.float32, device="cuda")
views = []
for k in range(nviews + 1):
    ro, rd, bd = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 2, **f32)
    teacher.k_gen_rays(hw, hw, focal, pose(2 * math.pi * k / (nviews + 1) + 0.2), 2.0, 6.0, False, ro, rd, bd)
    img = torch.empty(n, 3, **f32)
    teacher.render_dev(ro, rd, bd, rd, n, img)
    views.append((ro, rd, bd, img))
teacher.synchronize()
held = views[-1]
tr_o = torch.cat([v[0] for v in views[:-1]]); tr_d = torch.cat([v[1] for v in views[:-1]])
tr_b = torch.cat([v[2] for v in views[:-1]]); tr_l = torch.cat([v[3] for v in views[:-1]])
print("teacher image: mean %.3f std %.3f" % (float(held[3].mean()), float(held[3].std())))
teacher.close()

init = H.make_params(c, 5, bias_scale=0.0)
g = torch.Generator(device="cpu").manual_seed(7)
batches = [torch.randint(0, tr_o.shape[0], (B,), generator=g).cuda() for _ in range(steps)]
res = {}
for run in ("f32", "f32#2", "f16x3", "f16x3#2", "bf16", "bf16#2"):   # every dtype twice: the run-to-run spread (fp32 atomics order) is the noise floor
    dtype = run.split("#")[0]
    st = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=B, dtype=dtype, lrate=5e-4)))
    st.set_params(init)
    st.set_stream(torch.cuda.current_stream().cuda_stream)     # the batch gathers below run on torch's stream
    ev = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=n, dtype=dtype)))
    ev.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    curve = []
    for s, idx in enumerate(batches):
        st.train_step_dev(tr_o[idx].contiguous(), tr_d[idx].contiguous(), tr_b[idx].contiguous(), tr_d[idx].contiguous(),
                          tr_l[idx].contiguous(), B)
        if (s + 1) % max(1, steps // 4) == 0 or s + 1 == steps:
            st.synchronize()
            ev.set_params(st.get_params())
            out = torch.empty(n, 3, **f32)
            ev.render_dev(held[0], held[1], held[2], held[1], n, out)
            ev.synchronize()
            mse = float(((out - held[3]) ** 2).mean())
            curve.append((s + 1, -10 * math.log10(mse)))
    st.synchronize()
    res[run] = curve
    print(run, "train %.1f s" % (time.perf_counter() - t0), " ".join("step %d: %.2f dB" % x for x in curve))
    st.close(); ev.close()
print("held-out PSNR after %d steps: f32 %.2f / %.2f dB, f16x3 %.2f / %.2f dB, bf16 %.2f / %.2f dB" %
      (steps, res["f32"][-1][1], res["f32#2"][-1][1], res["f16x3"][-1][1], res["f16x3#2"][-1][1], res["bf16"][-1][1], res["bf16#2"][-1][1]))
