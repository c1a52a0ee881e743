"""Two ranks on ONE GPU over gloo, nothing of nerfhip involved: repeated all-reduce of a 4.24 MB device tensor on a
non-default stream.  Used to tell a gloo/torch artefact of GPU sharing from a fault of bench.py's rehearsal mode.
torchrun --nproc-per-node 2 --master-addr 127.0.0.1 scripts/gloo_probe.py [floats]"""
import faulthandler, os, sys, time
import torch
import torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1060360
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
g = torch.ones(n, device="cuda")
w = torch.randn(4096, 4096, device="cuda")
for i in range(5):
    for _ in range(20):
        w = (w @ w).clamp_(-1, 1)          # some device work in front of the collective
    g.fill_(1.0)
    dist.all_reduce(g)
    torch.cuda.synchronize()
    print(f"[probe rank {rank}] all_reduce {i} ok, g[0]={g[0].item()}", file=sys.stderr, flush=True)
dist.destroy_process_group()
