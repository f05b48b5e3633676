#!/bin/bash
# GPU box: rocprofv3 kernel statistics of one python tool.  bash tests/tools/kstats.sh <tag> <tool.py> [args...]  -> gpurun_out/<tag>_kernel_stats.csv
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 $R/"$@" > $O.log 2>&1
cp $O/p_kernel_stats.csv $R/gpurun_out/${T}_kernel_stats.csv
rm -rf $O
cut -d, -f1-4 $R/gpurun_out/${T}_kernel_stats.csv | head -${KSTATS_LINES:-12}
