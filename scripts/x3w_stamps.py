"""Diagnostic: per-entry cycles and per-workgroup spans of the split-f16 weight-gradient kernel, from a library built
with -DX3W_STAMPS (python scripts/x3w_stamps.py <library>), or of the fp32 one built with -DWGF_STAMPS (... <library> f32).  Prints cycles per chunk by product kind and the spread of
the workgroups' finishing times (what the equal-share cost model in train_impl.inc is fitted to)."""
import sys, os, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from nerf_amd import NerfBlock, NerfConfig
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
c = dict(H.CONFIGS["full"], dtype=dtype)
nT = 4096
tr = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nT)))
tr.init_params(1)
ro, rd, b, lab = H.make_rays(nT, 1)
dev = lambda a: torch.from_numpy(a).cuda()
o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
for _ in range(4):
    tr.train_step_dev(o, d, bb, d, l, nT)
tr.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
ent = np.zeros(8192 * 2, np.uint64); span = np.zeros(1024 * 2, np.uint64)
fn = {"f16x3": "nerfhip_debug_x3w_stamps", "f32": "nerfhip_debug_wgf_stamps", "bf16": "nerfhip_debug_wgb_stamps"}[dtype]
fn = getattr(lib, fn)
assert fn(ent.ctypes.data_as(C.c_void_p), span.ctypes.data_as(C.c_void_p)) == 0
ent = ent.reshape(-1, 2); span = span.reshape(-1, 2)[:256].astype(np.int64)
kinds = collections.defaultdict(list)
for cyc, info in ent:
    info = int(info)
    if info == 0: continue
    ar, br, nch, wg = info >> 48, (info >> 32) & 0xffff, (info >> 12) & 0xfffff, info & 0xfff
    kinds[(ar, br)].append((nch, int(cyc)))
for k, v in sorted(kinds.items()):
    v = np.array(v, float)
    if len(v) > 2 and np.ptp(v[:, 0]) > 0:
        A = np.stack([np.ones(len(v)), v[:, 0]], 1)
        (fix, per), *_ = np.linalg.lstsq(A, v[:, 1], rcond=None)
    else:
        fix, per = 0.0, v[:, 1].sum() / v[:, 0].sum()
    print(f"{k[0]:4d} x {k[1]:4d}: {len(v):4d} entries, {int(v[:, 0].sum()):7d} chunks, {v[:, 1].sum() / v[:, 0].sum():8.0f} cycles/chunk "
          f"(fit: {fix:8.0f} per entry + {per:7.0f} per chunk)")
t0 = span[:, 0].min()
dur = (span[:, 1] - span[:, 0]) / 100.0   # wall_clock64: 100 MHz
end = (span[:, 1] - t0) / 100.0
print(f"workgroup busy time us: min {dur.min():.0f} mean {dur.mean():.0f} max {dur.max():.0f}; last start {(span[:, 0].max() - t0) / 100.0:.0f}; "
      f"finish: min {end.min():.0f} mean {end.mean():.0f} max {end.max():.0f}")
# spread inside the largest kind, by workgroup and by XCD (workgroup id mod 8)
big = max(kinds, key=lambda k: sum(x[0] for x in kinds[k]))
rows = []
for cyc, info in ent:
    info = int(info)
    if info == 0: continue
    ar, br, nch, wg = info >> 48, (info >> 32) & 0xffff, (info >> 12) & 0xfffff, info & 0xfff
    if ar == big[0] and br == big[1] and nch > 100: rows.append((wg, nch, int(cyc) / nch))
rows = np.array(rows)
print(f"{big[0]}x{big[1]} cycles/chunk: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (rows[:, 2].min(), np.percentile(rows[:, 2], 10), np.median(rows[:, 2]), np.percentile(rows[:, 2], 90), rows[:, 2].max()))
for x in range(8):
    sel = rows[rows[:, 0].astype(int) % 8 == x]
    print(f"  xcd {x}: {len(sel):3d} entries, mean cycles/chunk {sel[:, 2].mean():.0f}; busy us of its workgroups: mean {dur[x::8].mean():.0f} max {dur[x::8].max():.0f}")
print("busy us by workgroup (every 8th):", " ".join(f"{v:.0f}" for v in dur[::8]))
# the slowest and fastest workgroups and what they ran
by_wg = collections.defaultdict(list)
for cyc, info in ent:
    info = int(info)
    if info == 0: continue
    by_wg[info & 0xfff].append((info >> 48, (info >> 32) & 0xffff, (info >> 12) & 0xfffff, int(cyc)))
order = np.argsort(dur)
for w in list(order[:4]) + list(order[-6:]):
    print(f"  wg {w:3d} busy {dur[w]:.0f} us:", "; ".join(f"{a}x{b} {n} chunks {c / 1e3:.0f}k cyc" for a, b, n, c in by_wg[int(w)]))
