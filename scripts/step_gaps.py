"""Where a training step's wall time goes: kernels vs gaps between them.
  python scripts/step_gaps.py run <f32|bf16> <rays> <steps>      (under rocprofv3 --kernel-trace)
  python scripts/step_gaps.py sum <kernel_trace.csv> <steps>     -> span of the last <steps> steps, kernel time, gap time
"""
import sys, os, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time, torch
    import helpers as H
    from nerf_amd import NerfBlock, NerfConfig
    dtype, nT, steps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    graph = len(sys.argv) > 5 and sys.argv[5] == "graph"
    c = dict(H.CONFIGS["full"], dtype=dtype)
    tr = NerfBlock(NerfConfig.from_dict(dict(c, maxRays=nT)))
    tr.init_params(1)
    if graph:
        tr.set_graph_mode(True)
    ro, rd, b, lab = H.make_rays(nT, 1)
    dev = lambda a: torch.from_numpy(a).cuda()
    o, d, bb, l = dev(ro), dev(rd), dev(b), dev(lab)
    for _ in range(5):
        tr.train_step_dev(o, d, bb, d, l, nT)
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step_dev(o, d, bb, d, l, nT)
    tr.synchronize()
    print(f"{dtype} rays {nT} graph {graph}: wall {1e3 * (time.perf_counter() - t0) / steps:.4f} ms/step")
else:
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "nh::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
    steps = int(sys.argv[3])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step ends with the last pack kernel behind adam_kernel; find the adam launches
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    first = adam[-steps - 1] + 1
    sel = rows[first:]
    # cut at the last step's end: everything up to the end of the file belongs to it (repack follows adam)
    span = int(sel[-1]["End_Timestamp"]) - int(rows[first - 1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    print(f"launches/step {len(sel) / steps:.1f}  span {span / steps / 1e6:.4f} ms/step  kernels {busy / steps / 1e6:.4f}  gaps {(span - busy) / steps / 1e6:.4f}")
    if len(sys.argv) > 4 and sys.argv[4] == "by":     # per-kernel share of a step (us per step, launches per step)
        by = {}
        for r in sel:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            t, n = by.get(k, (0, 0))
            by[k] = (t + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n + 1)
        for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:60s} {t / steps / 1e3:9.2f} us/step  {n / steps:5.1f} launches")
