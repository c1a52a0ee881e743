#!/bin/bash
# rocprofv3 over the HBM-bound kernels alone (scripts/hbm_kernels.py): kernel trace + separate PMC passes.
# usage: hbm_prof.sh <tag>   -> gpurun_out/<tag>_hbm_*.csv
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/scripts/hbm_kernels.py 10 > $R/gpurun_out/${TAG}_hbm_events.json
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hbm_${TAG}_trace -o p --output-format csv -- python3 $R/scripts/hbm_kernels.py 5 > /dev/null 2>&1
cp $R/gpurun_out/hbm_${TAG}_trace/p_kernel_stats.csv $R/gpurun_out/${TAG}_hbm_kernel_stats.csv
python3 -c "import sys; sys.path.insert(0, '$R'); from nerf_amd import _lib; import json; print(json.dumps({'nerfhip_version': _lib.load().nerfhip_version().decode(), 'tag': '${TAG}_hbm'}))" > $R/gpurun_out/${TAG}_hbm_pmc_meta.json
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/hbm_${TAG}_$name -o p --output-format csv -- python3 $R/scripts/hbm_kernels.py 5 > /dev/null 2>&1
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/hbm_${TAG}_$name/p_counter_collection.csv > $R/gpurun_out/${TAG}_hbm_pmc_$name.csv
  echo "pass $name done"
}
pass sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
pass fetchsize GRBM_GUI_ACTIVE FETCH_SIZE
pass writesize GRBM_GUI_ACTIVE WRITE_SIZE
cat $R/gpurun_out/${TAG}_hbm_events.json
