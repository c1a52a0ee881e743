import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, helpers as H, gpu_util as G
c=H.CONFIGS["full"]
n=160000
blk=G.make_block(c, max_rays=n)
blk.init_params(1)
ro,rd,b,_=H.make_rays(n,1)
o,d,bb=G.dev(ro),G.dev(rd),G.dev(b); out=G.empty(n,3)
torch.cuda.synchronize()
blk.render_dev(o,d,bb,d,n,out); blk.synchronize()
blk.set_profiling(True)
t=time.time()
for _ in range(3): blk.render_dev(o,d,bb,d,n,out)
blk.synchronize(); dt=(time.time()-t)/3
p=blk.get_profile()
print('render ms/frame',dt*1e3,'rays/s',n/dt)
print(p)
print('mlp fwd TFLOP/s', p['mlp_fwd_flop']/p['mlp_fwd_ms']/1e9)
print(out[:2])
