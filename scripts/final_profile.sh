#!/bin/bash
# round-end measurement set (run through gpurun): GPU tests, default bench, rocprofv3 kernel stats of the same bench
# command, PMC passes over the training steps (fp32, bf16), the HBM-bound kernels and the split-f16 render.
# usage: final_profile.sh <tag>      -> gpurun_out/<tag>_*  (copy what is judged into profiles/)
set -e
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1 || { tail -20 gpurun_out/${TAG}_gpu_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json
echo bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.log
cp $R/gpurun_out/${TAG}_stats/s_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
echo stats done
cd $R
bash scripts/pmc_train.sh bf16 ${TAG}_bf16
bash scripts/pmc_train.sh f32 ${TAG}_f32
bash scripts/pmc_train.sh f16x3 ${TAG}_f16x3train
bash scripts/hbm_prof.sh ${TAG}
bash scripts/x3_prof.sh f16x3 ${TAG}_f16x3
bash scripts/ns2_prof.sh ${TAG}_bf16render nerf_amd/libnerfhip.so
