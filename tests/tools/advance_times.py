#!/usr/bin/env python3
"""GPU box: device time (HIP events, elmk_profile_wrapper) of every wrapper of the reference's advance() order on one state.
python tests/tools/advance_times.py [cols] [tier] [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402
from elmkernels_amd import synth  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tier = sys.argv[2] if len(sys.argv) > 2 else "B"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
D, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
D.set_snow_age_tables(synth.snow_age_tables())
# the state each wrapper starts from: run the step up to it, snapshot what it changes, profile with restores
ALL = [n for n in D.fields if n != "err_flags"]
order = [("soil_temperature", lambda: st.kokkos_soil_temperature(D, 1800.0)), ("snow_hydrology", lambda: st.kokkos_snow_hydrology(D, 1800.0)),
         ("surface_fluxes", lambda: st.kokkos_surface_fluxes(D, 1800.0))]
st.kokkos_init_timestep(D)
for _ in range(3):
    D.restore_fields()
    st.timestep7(D, 1800.0)
for name, run in order:
    D.snapshot_fields(ALL if cols <= 2_000_000 else bench.SOIL_RESTORE)
    ms = D.profile_wrapper(st.WRAPPER_NAMES.index(name), 1800.0, steps)
    print(f"{name}: {ms:.3f} ms at {cols} columns tier {tier} ({cols / ms / 1e3:.1f} M columns/s)", flush=True)
    D.restore_fields()
    run()
# and the whole device part of advance() as one call, from the state after init_timestep
D2, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
D2.set_snow_age_tables(synth.snow_age_tables())
st.kokkos_init_timestep(D2)
D2.snapshot_fields(ALL if cols <= 2_000_000 else bench.SOIL_RESTORE)
D2.profile_wrapper(st.WRAPPER_NAMES.index("advance_physics"), 1800.0, 5)  # (the canopy scheduling hints settle)
ms = D2.profile_wrapper(st.WRAPPER_NAMES.index("advance_physics"), 1800.0, steps)
print(f"advance_physics (the seven fused + soil_temperature + snow_hydrology + surface_fluxes, one call): {ms:.3f} ms at {cols} columns tier {tier} ({cols / ms / 1e3:.1f} M gridcell-steps/s)", flush=True)
D2.close()
D.close()
