#!/bin/bash
# GPU box: the measurements DESIGN.md and profiles/ quote.  bash tests/tools/profile_round.sh <tag>  (outputs under gpurun_out/<tag>/)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-prof}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. HBM traffic per kernel (PMC, one counter per pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 B > $O/pmc_write.log 2>&1
python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic_pmc.json | tee $O/hbm_traffic.txt
cp $O/hbm_traffic_pmc.json $R/profiles/r01_hbm_traffic_pmc.json
# 2. the benchmark line (reads the table written above) and its rocprofv3 kernel statistics
cd $R
python3 bench.py > $O/bench_1M.json 2> $O/bench_1M.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_1M_under_rocprof.json 2> $O/kt.log
cp $O/kt/p_kernel_stats.csv $O/bench_1M_rocprof_kernel_stats.csv
cd $R
python3 bench.py --cols 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_10M.json 2> $O/bench_10M.err
rm -rf $O/pmc_fetch $O/pmc_write $O/kt
echo done
