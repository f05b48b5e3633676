#!/bin/bash
# GPU box: HBM traffic per kernel (two rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE) of one configuration of one library build.
#   bash tests/tools/pmc_quick.sh <tag> <mode: timestep7|fused|soil|snow> <tier> [cols] [lib.so]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; MODE=$2; TIER=$3; COLS=${4:-1000000}; LIB=$5
O=$R/gpurun_out/$T
mkdir -p $O
[ -n "$LIB" ] && export ELMK_LIBRARY=$R/$LIB
N=$(basename ${LIB:-libelmk} .so)_${MODE}_tier${TIER}_${COLS}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py $COLS $TIER $MODE > $O/pf_$N.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 $R/tests/tools/traffic_probe.py $COLS $TIER $MODE > $O/pw_$N.log 2>&1
python3 $R/tests/tools/make_traffic_json.py $O/pf $O/pw $O/traffic_$N.json $COLS $TIER | tee $O/traffic_$N.txt
python3 - $O/traffic_$N.json $COLS <<'PY' | tee -a $O/traffic_$N.txt
import json, sys
t = json.load(open(sys.argv[1])); n = float(sys.argv[2])
for k, v in t["kernels"].items():
    if k in ("elmk::k_copy",): continue
    print(f"  {k:30s} fetch {2*v['FETCH_SIZE_KiB']*1024/n:8.1f} B/col  write {v['WRITE_SIZE_KiB']*1024/n:8.1f} B/col  total {v['hbm_bytes_per_launch']/n:8.1f}")
PY
rm -rf $O/pf $O/pw
