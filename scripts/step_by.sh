#!/bin/bash
# per-kernel share of a training step from the rocprofv3 kernel trace: step_by.sh <dtype> <rays>
set -e
R=$GRAFT_REPO_ROOT
DT=$1; N=$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace -d $R/gpurun_out/by_${DT}_${N} -o p --output-format csv -- python3 $R/scripts/step_gaps.py run $DT $N 50 > /dev/null 2>&1
python3 $R/scripts/step_gaps.py sum $R/gpurun_out/by_${DT}_${N}/p_kernel_trace.csv 40 by
