#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the same timing script for several builds of the library (A/B runs).
#   bash tests/tools/ab.sh <tag> "<kernel-name regex>" <lib1.so> [lib2.so ...]   (KT_* / AB_COLS / AB_TIERS from the environment)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; PAT=$2; shift 2
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  for tier in ${AB_TIERS:-A B}; do
    O=$R/gpurun_out/$T/${name}_$tier
    ELMK_LIBRARY=$R/$lib rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 $R/tests/tools/kernel_times.py ${AB_COLS:-1000000} $tier ${AB_STEPS:-10} > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
    echo "== $name tier $tier: $(grep '^tier' $O.log | cut -d'|' -f1)"
    python3 - "$O/p_kernel_stats.csv" "$PAT" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Name"])]
for r in rows:
    print(f"   {r['Name'].split('(')[0][:44]:46s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
    cp $O/p_kernel_stats.csv $R/gpurun_out/$T/${name}_${tier}_kernel_stats.csv
    rm -rf $O
  done
done
