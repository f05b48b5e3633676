#!/usr/bin/env python3
"""GPU box: device time of elmk_soil_temperature (BASELINE config 3: the soil-column vertical solve).
python tests/tools/soil_temp_time.py [cols] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D, _ = bench.build_state(cols, 0, "B", 0x5EEDE1A0)
D.snapshot_fields(["t_soisno", "h2osoi_ice", "h2osoi_liq", "t_h2osfc", "h2osfc", "h2osno", "snow_depth", "int_snow", "t_grnd"])
st.timestep7(D, 1800.0)
D.snapshot_fields(["t_soisno", "h2osoi_ice", "h2osoi_liq", "t_h2osfc", "h2osfc", "h2osno", "snow_depth", "int_snow", "t_grnd"])
for _ in range(2):
    D.restore_fields()
    st.kokkos_soil_temperature(D, 1800.0)
D.sync()
t_restore = time.perf_counter()
for _ in range(reps):
    D.restore_fields()
D.sync()
t_restore = (time.perf_counter() - t_restore) / reps
t0 = time.perf_counter()
for _ in range(reps):
    D.restore_fields()
    st.kokkos_soil_temperature(D, 1800.0)
D.sync()
dt = (time.perf_counter() - t0) / reps - t_restore
ALGO = 2600  # bytes per column (SURVEY 8(d) estimate for this row)
print(f"soil_temperature {cols} columns: {dt * 1e3:.3f} ms per call ({cols / dt / 1e6:.1f} M columns/s, "
      f"{ALGO * cols / dt / 1e9:.0f} GB/s of ~{ALGO} B/column = {ALGO * cols / dt / 8e12 * 100:.1f} % of 8 TB/s); restore {t_restore * 1e3:.3f} ms")
flags, first = D.error_summary()
print("error flags", flags)
D.close()
