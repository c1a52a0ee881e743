#!/bin/bash
# timing experiment: fp32 weight-gradient kernel with one product kind at a time (gradients wrong by construction)
: > gpurun_out/kinds.txt
for k in 0 1 9 10 11 12 13 14 22 23 24; do
  NERFHIP_DEBUG_WGRAD_ONLY=$k timeout -k 10 120 python bench.py $EXTRA --no-cpu-baseline --render-frames 1 --no-bf16-train --no-bf16-render --steps 5 --warmup 1 > gpurun_out/bk.json 2> gpurun_out/bk.err || exit 1
  grep "wgrad only" gpurun_out/bk.err | head -1 >> gpurun_out/kinds.txt
  python -c "
import json; d=json.load(open('gpurun_out/bk.json')); print('   wgrad_ms', d['roofline']['ms_per_step']['mlp_wgrad'])" >> gpurun_out/kinds.txt
done
