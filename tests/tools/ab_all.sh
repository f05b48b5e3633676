#!/bin/bash
# GPU box: interleaved A/B of whole library builds - per-wrapper step and fused step at 1 M columns (rocprofv3 kernel averages,
# tests/tools/ab.sh) and the soil-column solve at 10 M columns (tests/tools/soil_temp_time.py).
#   bash tests/tools/ab_all.sh <tag> <lib1.so> <lib2.so> ...      (AB_TIERS, AB_STEPS, AB_SOIL=0 to skip the soil solve)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; shift
mkdir -p $R/gpurun_out/$T
{
echo "# per-wrapper step, 1 M columns"
bash $R/tests/tools/ab.sh $T/w "elmk::k_" "$@"
echo "# fused step, 1 M columns"
KT_FUSED=1 bash $R/tests/tools/ab.sh $T/f "elmk::k_" "$@"
if [ "${AB_SOIL:-1}" != "0" ]; then
  echo "# soil-column solve, 10 M columns"
  for lib in "$@"; do
    echo "== $(basename $lib .so): $(ELMK_LIBRARY=$R/$lib python3 $R/tests/tools/soil_temp_time.py 10000000 10 | head -1)"
  done
fi
} 2>&1 | tee $R/gpurun_out/$T/summary.txt
