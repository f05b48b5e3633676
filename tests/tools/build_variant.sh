#!/bin/bash
# Build container: a development build of libelmk with ONE kernel file replaced or re-flagged, selected on the GPU box through
# ELMK_LIBRARY.  bash tests/tools/build_variant.sh <out.so> <kernel.hip | path/to/other/version.hip> [-DFLAG ...]
# (the file's base name decides which object of the product build it replaces)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
OUT=$1; SRC=$2; shift 2
cd $R/elmkernels_amd/csrc
make -s -j8
[ -f "$SRC" ] || SRC=$R/elmkernels_amd/csrc/$SRC
B=$(basename $SRC .hip)
T=$(mktemp -d /tmp/variant.XXXX)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -Wall -Wno-unused-function -I../../include -I. "$@" -c $SRC -o $T/$B.o 2> >(grep -v "argument unused" >&2)
OBJS=""
for o in build/*.o; do
  if [ "$(basename $o .o)" = "$B" ]; then OBJS="$OBJS $T/$B.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $OBJS 2> >(grep -v "argument unused" >&2)
rm -rf $T
