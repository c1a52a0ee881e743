#!/bin/bash
# PMC passes over a few train steps (scripts/prof_train.py): usage pmc_train.sh <f32|bf16> <tag>
# One counter block set per pass (rocprofv3 does not split counters over passes); summaries land in gpurun_out/.
set -e
DT=$1; TAG=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_${TAG}_$name -o p --output-format csv -- python3 $R/scripts/prof_train.py $DT > /dev/null 2>&1
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name/p_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_$name.csv
  echo "pass $name done"
}
# which library these counters belong to: bench.py only quotes a summary whose string equals the loaded library's
python3 -c "import sys; sys.path.insert(0, '$R'); from nerf_amd import _lib; import json; print(json.dumps({'nerfhip_version': _lib.load().nerfhip_version().decode(), 'tag': '$TAG', 'dtype': '$DT', 'steps': 3}))" > $R/gpurun_out/${TAG}_pmc_meta.json
pass sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass fetch GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
pass fetchsize GRBM_GUI_ACTIVE FETCH_SIZE
pass writesize GRBM_GUI_ACTIVE WRITE_SIZE
