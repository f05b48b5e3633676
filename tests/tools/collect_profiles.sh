#!/bin/bash
# In the build container, after `gpurun -- bash tests/tools/profile_round.sh <tag>`: only gpurun_out/ travels back from the GPU
# box, so copy the files to be committed from gpurun_out/<tag>/ into profiles/ with the round prefix.
# bash tests/tools/collect_profiles.sh <tag>
set -e
T=${1:-prof}
P=${PROFILE_TAG:-r04}
O=gpurun_out/$T
for f in hbm_traffic_pmc_tierA.json hbm_traffic_pmc_tierB.json hbm_traffic_pmc_fused_tierA.json hbm_traffic_pmc_fused_tierB.json \
         hbm_traffic_pmc_soil_tierB.json hbm_traffic_pmc_fused_f32_tierA.json hbm_traffic_pmc_fused_f32_tierB.json compute_pmc_tierA.json compute_pmc_tierB.json compute_pmc_tierA.txt compute_pmc_tierB.txt bench_1M.json bench_1M_rocprof_kernel_stats.csv bench_1M_fused_rocprof_kernel_stats.csv \
         hbm_traffic_pmc_10M_tierA.json hbm_traffic_pmc_10M_tierB.json hbm_traffic_pmc_10M_fused_tierA.json hbm_traffic_pmc_10M_fused_tierB.json \
         hbm_traffic_pmc_10M_fused_f32_tierA.json hbm_traffic_pmc_10M_fused_f32_tierB.json bench_10M_rocprof_kernel_stats.csv bench_10M_fused_rocprof_kernel_stats.csv \
         compute_pmc_fused_tierA.json compute_pmc_fused_tierB.json compute_pmc_10M_tierA.json compute_pmc_10M_tierB.json compute_pmc_10M_fused_tierA.json compute_pmc_10M_fused_tierB.json \
         compute_pmc_fused_tierA.txt compute_pmc_fused_tierB.txt compute_pmc_10M_tierA.txt compute_pmc_10M_tierB.txt compute_pmc_10M_fused_tierA.txt compute_pmc_10M_fused_tierB.txt \
         bench_soil_10M.json bench_soil_1M.json advance_times_1M.txt bench_2ranks_on_1gpu.json device_info.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/${P}_$f
done
ls profiles | grep "^${P}_"
# the SQ counter summary and step timelines of tests/tools/pmc_round.sh, if that was run too (gpurun_out/pmc_r3/)
for f in pmc_counters tl_per_wrapper_tierA tl_fused_tierA tl_fused_tierB tl_advance_tierB; do
  [ -f gpurun_out/pmc_r3/$f.txt ] && cp gpurun_out/pmc_r3/$f.txt profiles/${P}_$f.txt
done
