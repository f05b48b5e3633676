#!/bin/bash
# GPU box: hardware counters of the kernels one python tool launches, one rocprofv3 pass per counter group (groups are
# separated by ';' in $PMC_GROUPS, counters inside a group by spaces; at most 8 SQ counters per group).
# PMC_GROUPS="SQ_WAVE_CYCLES SQ_WAIT_ANY;SQ_IFETCH" bash tests/tools/pmc_groups.sh <tag> <kernel-substring> <tool.py> [tool args...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; K=$2; shift 2
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra GROUPS_ <<< "$PMC_GROUPS"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/"$@" > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; echo "group $i failed"; continue; }
  python3 $R/tests/tools/pmc_summary.py $O/g$i "$K"
done | tee $R/gpurun_out/$T.txt
rm -rf $O
