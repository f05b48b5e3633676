#!/usr/bin/env python3
"""GPU box, under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: a calibration copy of known size with the kernels' own
access shape (8 bytes per lane), then a few timesteps.  python tests/tools/traffic_probe.py [cols] [tier]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tier = sys.argv[2] if len(sys.argv) > 2 else "B"
mode = sys.argv[3] if len(sys.argv) > 3 else "timestep7"  # timestep7 | fused | soil | snow
D, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
print("copy GB/s (1 GiB buffers, 8 B/lane):", D.copy_bandwidth(1 << 30, 5))
if mode == "soil":
    st.timestep7(D, 1800.0)
    D.snapshot_fields(bench.SOIL_RESTORE)
    for _ in range(3):
        D.restore_fields()
        st.kokkos_soil_temperature(D, 1800.0)
elif mode == "snow":
    from elmkernels_amd import synth

    D.set_snow_age_tables(synth.snow_age_tables())
    st.kokkos_init_timestep(D)
    st.timestep7(D, 1800.0)
    st.kokkos_soil_temperature(D, 1800.0)
    D.snapshot_fields([n for n in D.fields if n != "err_flags"])
    for _ in range(3):
        D.restore_fields()
        st.kokkos_snow_hydrology(D, 1800.0)
else:
    adv = st.timestep7_fused if mode == "fused" else st.timestep7
    for _ in range(8):  # (the canopy_fluxes scheduling hints settle over a few steps)
        D.restore_fields()
        adv(D, 1800.0)
D.sync()
print("done")
