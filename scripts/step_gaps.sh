#!/bin/bash
# usage: step_gaps.sh <dtype> <rays> [graph]
set -e
R=$GRAFT_REPO_ROOT
DT=$1; N=$2; G=$3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/scripts/step_gaps.py run $DT $N 50 $G
timeout -s KILL 120 rocprofv3 --kernel-trace -d $R/gpurun_out/gaps_${DT}_${N}$G -o p --output-format csv -- python3 $R/scripts/step_gaps.py run $DT $N 50 $G > /dev/null 2>&1
python3 $R/scripts/step_gaps.py sum $R/gpurun_out/gaps_${DT}_${N}$G/p_kernel_trace.csv 40
