#!/bin/bash
# GPU box: the measurements DESIGN.md and profiles/ quote.  bash tests/tools/profile_round.sh <tag> [pmc|pmc10|sq|bench|rest|all]
# (outputs under gpurun_out/<tag>/; tests/tools/collect_profiles.sh copies the files to commit into profiles/ with the round
# prefix.  One gpurun call is limited to 20 minutes, so a round is four calls - pmc, pmc10, then bench, then rest - with a
# collect_profiles.sh after the first: the benchmark reads the PMC tables from profiles/.)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-prof}
STAGE=${2:-all}
P=${PROFILE_TAG:-r04}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $STAGE = pmc ] || [ $STAGE = all ]; then
# 1. HBM traffic per kernel (PMC, one counter per pass): the per-wrapper step and the fused step on both synthetic tiers,
#    and the soil-column solve
for spec in "A timestep7 tierA" "B timestep7 tierB" "A fused fused_tierA" "B fused fused_tierB" "B soil soil_tierB"; do
  set -- $spec; tier=$1; mode=$2; name=$3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier $mode > $O/pmc_fetch_$name.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier $mode > $O/pmc_write_$name.log 2>&1
  python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic_pmc_$name.json 1000000 $tier | tee $O/hbm_traffic_$name.txt
  cp $O/hbm_traffic_pmc_$name.json $R/profiles/${P}_hbm_traffic_pmc_$name.json
  rm -rf $O/pmc_fetch $O/pmc_write
done
# 1a. the fp32-state build (BASELINE config 5's report-only variant): traffic of its fused step
export ELMK_LIBRARY=$R/elmkernels_amd/libelmk_f32.so
for tier in A B; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier fused > $O/pmc_fetch_f32_$tier.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier fused > $O/pmc_write_f32_$tier.log 2>&1
  python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic_pmc_fused_f32_tier$tier.json 1000000 $tier | tee $O/hbm_traffic_fused_f32_tier$tier.txt
  cp $O/hbm_traffic_pmc_fused_f32_tier$tier.json $R/profiles/${P}_hbm_traffic_pmc_fused_f32_tier$tier.json
  rm -rf $O/pmc_fetch $O/pmc_write
done
unset ELMK_LIBRARY
# 1b. the compute side (SQ counters, one group per pass): VALU busy, VALU lane utilisation, fp64 instruction mix of every kernel
for tier in A B; do
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_lane -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier timestep7 > $O/pmc_lane_$tier.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/pmc_f64 -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 1000000 $tier timestep7 > $O/pmc_f64_$tier.log 2>&1
  python3 $R/tests/tools/make_compute_json.py $O/compute_pmc_tier$tier.json 1000000 $tier $O/pmc_lane $O/pmc_f64 | tee $O/compute_pmc_tier$tier.txt
  cp $O/compute_pmc_tier$tier.json $R/profiles/${P}_compute_pmc_tier$tier.json
  rm -rf $O/pmc_lane $O/pmc_f64
done
fi
if [ $STAGE = pmc10 ] || [ $STAGE = all ]; then
# 1c. the same traffic passes AT THE NORTH-STAR SIZE, 10 M columns (VERDICT r03 missing #4): per-wrapper and fused step, both tiers,
#     and the fp32-state build's fused step; plus the rocprofv3 kernel statistics of both steps at that size
for spec in "A timestep7 10M_tierA" "B timestep7 10M_tierB" "A fused 10M_fused_tierA" "B fused 10M_fused_tierB"; do
  set -- $spec; tier=$1; mode=$2; name=$3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 10000000 $tier $mode > $O/pmc_fetch_$name.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 10000000 $tier $mode > $O/pmc_write_$name.log 2>&1
  python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic_pmc_$name.json 10000000 $tier | tee $O/hbm_traffic_$name.txt
  cp $O/hbm_traffic_pmc_$name.json $R/profiles/${P}_hbm_traffic_pmc_$name.json
  rm -rf $O/pmc_fetch $O/pmc_write
done
export ELMK_LIBRARY=$R/elmkernels_amd/libelmk_f32.so
for tier in A B; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 10000000 $tier fused > $O/pmc_fetch_10M_f32_$tier.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py 10000000 $tier fused > $O/pmc_write_10M_f32_$tier.log 2>&1
  python3 $R/tests/tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic_pmc_10M_fused_f32_tier$tier.json 10000000 $tier | tee $O/hbm_traffic_10M_fused_f32_tier$tier.txt
  cp $O/hbm_traffic_pmc_10M_fused_f32_tier$tier.json $R/profiles/${P}_hbm_traffic_pmc_10M_fused_f32_tier$tier.json
  rm -rf $O/pmc_fetch $O/pmc_write
done
unset ELMK_LIBRARY
for mode in "" "--fused"; do
  n=bench_10M${mode:+_fused}
  rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py $mode --cols 10000000 --steps 5 --warmup 2 --no-cpu-baseline --no-other-tier --no-north-star --no-state-f32 --no-two-blocks > $O/${n}_under_rocprof.json 2> $O/kt10.log
  cp $O/kt/p_kernel_stats.csv $O/${n}_rocprof_kernel_stats.csv
  cp $O/${n}_rocprof_kernel_stats.csv $R/profiles/${P}_${n}_rocprof_kernel_stats.csv
  rm -rf $O/kt
done
fi
if [ $STAGE = sq ] || [ $STAGE = all ]; then
# 1d. the compute side beyond the per-wrapper step at 1 M columns: the fused step's kernels at 1 M, and both steps at 10 M (VERDICT r03 missing #4)
for spec in "1000000 fused fused_" "10000000 timestep7 10M_" "10000000 fused 10M_fused_"; do
  set -- $spec; cols=$1; mode=$2; pre=$3
  for tier in A B; do
    rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_lane -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py $cols $tier $mode > $O/pmc_lane_${pre}$tier.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/pmc_f64 -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py $cols $tier $mode > $O/pmc_f64_${pre}$tier.log 2>&1
    python3 $R/tests/tools/make_compute_json.py $O/compute_pmc_${pre}tier$tier.json $cols $tier $O/pmc_lane $O/pmc_f64 | tee $O/compute_pmc_${pre}tier$tier.txt
    cp $O/compute_pmc_${pre}tier$tier.json $R/profiles/${P}_compute_pmc_${pre}tier$tier.json
    rm -rf $O/pmc_lane $O/pmc_f64
  done
done
fi
if [ $STAGE = bench ] || [ $STAGE = all ]; then
# 2. the benchmark line (reads the tables written above) and its rocprofv3 kernel statistics
cd $R
python3 bench.py --state-f32 > $O/bench_1M.json 2> $O/bench_1M.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-tier --no-north-star --no-two-blocks > $O/bench_1M_under_rocprof.json 2> $O/kt.log
cp $O/kt/p_kernel_stats.csv $O/bench_1M_rocprof_kernel_stats.csv
rm -rf $O/kt
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --fused --no-cpu-baseline --no-other-tier --no-north-star --no-two-blocks > $O/bench_1M_fused_under_rocprof.json 2> $O/ktf.log
cp $O/kt/p_kernel_stats.csv $O/bench_1M_fused_rocprof_kernel_stats.csv
rm -rf $O/kt
for f in bench_1M.json bench_1M_rocprof_kernel_stats.csv bench_1M_fused_rocprof_kernel_stats.csv; do cp $O/$f $R/profiles/${P}_$f; done
fi
if [ $STAGE = rest ] || [ $STAGE = all ]; then
cd $R
# 3. BASELINE config 3: the soil-column vertical solve
python3 bench.py --workload soil_temperature --cols 10000000 --steps 5 --warmup 2 --tier B > $O/bench_soil_10M.json 2> $O/bench_soil_10M.err
python3 bench.py --workload soil_temperature --no-cpu-baseline --tier B > $O/bench_soil_1M.json 2> $O/bench_soil_1M.err
# 4. the rest of advance() and the multi-rank rehearsal (two ranks sharing this one GPU, gloo for the barrier)
python3 tests/tools/advance_times.py 1000000 B 5 > $O/advance_times_1M.txt 2>&1
python3 bench.py --gpus 2 --cols 500000 --no-cpu-baseline > $O/bench_2ranks_on_1gpu.json 2> $O/bench_2ranks.err
# 5. the box itself (record what the roofline denominators refer to)
{ rocminfo | grep -E "Marketing Name|Name: +gfx|Compute Unit|Max Clock Freq|Wavefront Size|Workgroup Max Size:|Size: +[0-9]+\(0x[0-9a-f]+\) KB" | sort | uniq -c | sort -rn | head -40; echo; rocm-smi --showmeminfo vram --showclocks --showproductname 2>/dev/null | grep -v "^$" | head -60; } > $O/device_info.txt 2>&1 || true
for f in bench_soil_10M.json bench_soil_1M.json advance_times_1M.txt bench_2ranks_on_1gpu.json device_info.txt; do cp $O/$f $R/profiles/${P}_$f; done
fi
echo done
