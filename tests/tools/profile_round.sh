#!/bin/bash
# GPU box: the measurements DESIGN.md and profiles/ quote.  bash tests/tools/profile_round.sh <tag>  (outputs under gpurun_out/<tag>/)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-prof}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. HBM traffic per kernel (PMC, one counter per pass), both synthetic tiers
for tier in A B; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch$tier -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier > $O/pmc_fetch$tier.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write$tier -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier > $O/pmc_write$tier.log 2>&1
  python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch$tier $O/pmc_write$tier $O/hbm_traffic_pmc_tier$tier.json 1000000 $tier | tee $O/hbm_traffic_tier$tier.txt
  cp $O/hbm_traffic_pmc_tier$tier.json $R/profiles/r01_hbm_traffic_pmc_tier$tier.json
  rm -rf $O/pmc_fetch$tier $O/pmc_write$tier
done
# 2. the benchmark line (reads the tables written above) and its rocprofv3 kernel statistics
cd $R
python3 bench.py > $O/bench_1M.json 2> $O/bench_1M.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-tier > $O/bench_1M_under_rocprof.json 2> $O/kt.log
cp $O/kt/p_kernel_stats.csv $O/bench_1M_rocprof_kernel_stats.csv
rm -rf $O/kt
cd $R
python3 bench.py --cols 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_10M.json 2> $O/bench_10M.err
# 3. BASELINE config 3: the soil-column vertical solve
python3 bench.py --workload soil_temperature --cols 10000000 --steps 5 --warmup 2 --tier B > $O/bench_soil_10M.json 2> $O/bench_soil_10M.err
python3 bench.py --workload soil_temperature --no-cpu-baseline --tier B > $O/bench_soil_1M.json 2> $O/bench_soil_1M.err
echo done
