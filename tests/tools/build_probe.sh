#!/bin/bash
# Build container: a development build of libelmk with extra -D flags for k_canopy_fluxes.hip (CF_PROBE=5 ...), selected on
# the GPU box through ELMK_LIBRARY.  bash tests/tools/build_probe.sh <out.so> [-DCF_PROBE=5 ...]
set -e
R=$(cd $(dirname $0)/../.. && pwd)
OUT=$1; shift
cd $R/elmkernels_amd/csrc
make -s -j8
mkdir -p /tmp/probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -Wall -Wno-unused-function -I../../include -I. "$@" -c k_canopy_fluxes.hip -o /tmp/probe/k_canopy_fluxes.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT /tmp/probe/k_canopy_fluxes.o build/k_water_energy.o build/k_albedo_snicar.o build/k_soil_temperature.o build/k_snow_hydrology.o build/k_init_state.o build/k_surface_fluxes.o build/k_forcing.o build/k_util.o build/elmk_api.o
