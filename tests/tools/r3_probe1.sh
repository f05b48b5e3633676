#!/bin/bash
# GPU box, round 3 measurement batch 1: instruction issue costs, copy shapes, the probe timeline of k_cf_iterate (sections,
# root-find lane utilisation) and the lane-utilisation PMC pass.  bash tests/tools/r3_probe1.sh  -> gpurun_out/r3p1/
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3p1
mkdir -p $O
cd $R
echo "== valu_cost"; ./tests/tools/ubench/bin/valu_cost > $O/valu_cost.txt 2>&1; cat $O/valu_cost.txt
echo "== fp64_issue"; ./tests/tools/ubench/bin/fp64_issue > $O/fp64_issue.txt 2>&1; tail -4 $O/fp64_issue.txt
echo "== copy shapes"
python3 - > $O/copy_shapes.txt 2>&1 <<'PY'
import bench
D, _ = bench.build_state(100000, 0, "A", 0x5EEDE1A0)
for nb in (1 << 28, 1 << 30, 1 << 32):
    print("bytes", nb, {s: round(D.copy_bandwidth(nb, 10, s), 1) for s in (0, 1, 2, 3)})
D.close()
PY
cat $O/copy_shapes.txt
echo "== probe timeline"
ELMK_LIBRARY=$R/elmkernels_amd/libelmk_probe5.so CF_TIER=A python3 tests/tools/cf_timeline.py 1000000 > $O/cft_a.txt 2>&1; tail -12 $O/cft_a.txt
ELMK_LIBRARY=$R/elmkernels_amd/libelmk_probe5.so CF_TIER=B python3 tests/tools/cf_timeline.py 1000000 > $O/cft_b.txt 2>&1; tail -12 $O/cft_b.txt
echo "== lane utilisation PMC"
export PMC_GROUPS="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
bash tests/tools/pmc_groups.sh r3p1/lane_cf_A k_cf_iterate tests/tools/traffic_probe.py 1000000 A timestep7
bash tests/tools/pmc_groups.sh r3p1/lane_B k_ tests/tools/traffic_probe.py 1000000 B timestep7
echo done
