#!/bin/bash
# GPU box: rocprofv3 kernel trace (start / end of every kernel) of one python tool -> gpurun_out/<tag>_kernel_trace.csv
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o p --output-format csv -- python3 $R/"$@" > $O.log 2>&1
cp $O/p_kernel_trace.csv $R/gpurun_out/${T}_kernel_trace.csv
rm -rf $O
