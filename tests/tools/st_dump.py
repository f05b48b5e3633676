#!/usr/bin/env python3
"""GPU box, development: run the seven wrappers + soil_temperature and dump its outputs.  python tests/tools/st_dump.py out.npz [cols]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
D, _ = bench.build_state(cols, 0, "B", 0x5EEDE1A0)
st.timestep7(D, 1800.0)
st.kokkos_soil_temperature(D, 1800.0)
D.sync()
names = ["t_soisno", "t_h2osfc", "t_grnd", "h2osoi_ice", "h2osoi_liq", "h2osfc", "h2osno", "int_snow", "snow_depth", "fact",
         "sabg_chk", "xmf", "xmf_h2osfc", "qflx_h2osfc_ice", "eflx_h2osfc_snow", "qflx_snofrz", "qflx_snow_melt",
         "qflx_snomelt", "eflx_snomelt", "qflx_snofrz_lyr", "imelt"]
np.savez(sys.argv[1], **{k: D.download(k) for k in names})
