#!/bin/bash
# record of the experiment that fixed the bf16 weight-gradient cost model (train_bf16.inc: chunk_cost); it relied on a
# temporary NERFHIP_DEBUG_WB_COST=k0,k1 knob: best (4, 0.25) 1.877 ms vs (0,0) 2.013 ms.
: > gpurun_out/wbcost.txt
for kk in 0,0 2,0 4,0 8,0 0,0.25 2,0.25 4,0.25 4,0.5 8,0.5 16,0; do
  NERFHIP_DEBUG_WB_COST=$kk timeout -k 10 120 python bench.py --dtype bf16 --no-cpu-baseline --render-frames 1 --no-bf16-train --no-bf16-render --steps 10 --warmup 2 > gpurun_out/bk.json 2> gpurun_out/bk.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/bk.json')); print('$kk wgrad_ms', d['roofline']['ms_per_step']['mlp_wgrad'])" >> gpurun_out/wbcost.txt
done
