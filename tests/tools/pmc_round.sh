#!/bin/bash
# GPU box: the SQ counter passes DESIGN.md quotes (VALU-busy fractions, wave wait fractions) for the four kernels that are not
# bandwidth-bound, and the kernel timelines of one step.  bash tests/tools/pmc_round.sh <tag>  -> gpurun_out/<tag>/
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-pmc_r3}
O=$R/gpurun_out/$T
mkdir -p $O
export PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS;GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
cd $R
bash tests/tools/pmc_groups.sh $T/cf k_cf_iterate tests/tools/traffic_probe.py 1000000 A timestep7 > /dev/null
bash tests/tools/pmc_groups.sh $T/alb k_alb_snicar tests/tools/traffic_probe.py 1000000 A timestep7 > /dev/null
bash tests/tools/pmc_groups.sh $T/soil k_soil_temperature tests/tools/traffic_probe.py 1000000 B soil > /dev/null
bash tests/tools/pmc_groups.sh $T/snow k_snow_hydrology tests/tools/traffic_probe.py 1000000 B snow > /dev/null
{
  echo "# rocprofv3 --pmc passes (one group per pass), mean per launch, 1 M columns; SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* are in"
  echo "# units of 4 cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = value / 8)."
  echo "# VALU-busy = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); wait fraction = SQ_WAIT_ANY / SQ_WAVE_CYCLES."
  for k in cf alb soil snow; do echo "## $k"; cat $R/gpurun_out/$T/$k.txt; done
} > $O/pmc_counters.txt
cd /tmp && export TMPDIR=/tmp
for spec in "A x tl_per_wrapper_tierA" "A fused tl_fused_tierA" "B fused tl_fused_tierB" "B advance tl_advance_tierB"; do
  set -- $spec
  rocprofv3 --kernel-trace -d $O/tl -o p --output-format csv -- python3 $R/tests/tools/step_timeline.py run 1000000 $1 $2 > /dev/null 2>&1
  python3 $R/tests/tools/step_timeline.py parse $O/tl > $O/$3.txt
  rm -rf $O/tl
done
echo done
