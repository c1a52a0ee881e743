#!/bin/bash
# PMC pass over the render kernels of one dtype: usage x3_prof.sh <dtype> <tag> [library]
set -e
DT=$1; TAG=$2; LIB=$3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/x3_${TAG} -o p --output-format csv -- python3 $R/scripts/bench_render.py $DT 2 $LIB > $R/gpurun_out/x3_${TAG}.txt 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/x3_${TAG}/p_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_sq.csv
head -2 $R/gpurun_out/${TAG}_pmc_sq.csv | tail -1 | cut -d, -f1-12
