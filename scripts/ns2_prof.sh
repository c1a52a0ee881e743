#!/bin/bash
# PMC pass over the bf16 render kernel of an experimental library (round 3: one vs two sample tiles per wave):
# usage ns2_prof.sh <tag> <library>   -> gpurun_out/<tag>_pmc_sq.csv
set -e
TAG=$1; LIB=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $R/gpurun_out/ns2_${TAG} -o p --output-format csv -- python3 $R/scripts/bench_render.py bf16 2 $R/$LIB > $R/gpurun_out/ns2_${TAG}.txt 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/ns2_${TAG}/p_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_sq.csv
head -2 $R/gpurun_out/${TAG}_pmc_sq.csv | cut -d, -f1-13
