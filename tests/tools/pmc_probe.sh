#!/bin/bash
# GPU box: hardware counters of the kernels one python tool launches, one rocprofv3 pass per counter group.
# bash tests/tools/pmc_probe.sh <tag> <kernel-substring> <tool.py> [tool args...]     (output: gpurun_out/<tag>.txt)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; K=$2; shift 2
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/"$@" > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; echo "group $i failed"; continue; }
  python3 $R/tests/tools/pmc_summary.py $O/g$i "$K"
done | tee $R/gpurun_out/$T.txt
rm -rf $O
