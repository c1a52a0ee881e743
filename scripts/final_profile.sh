#!/bin/bash
# round-end measurement set: GPU tests, default bench, rocprofv3 kernel stats of the same bench command, PMC passes
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/final_gpu_tests.log 2>&1 || { tail -20 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json
echo bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final_bench_under_rocprof.json 2> $R/gpurun_out/final_rocprof.log
echo stats done
cd $R
bash scripts/pmc_train.sh bf16 final_bf16
bash scripts/pmc_train.sh f32 final_f32
