#!/usr/bin/env python3
"""GPU box, development: run 2 steps and dump canopy_fluxes outputs + trip counts.  python tests/tools/cf_dump.py out.npz [cols]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
D, _ = bench.build_state(cols, 0, "B", 0x5EEDE1A0)
for _ in range(2):
    D.restore_fields()
    st.timestep7(D, 1800.0)
D.sync()
np.savez(sys.argv[1], trips=D.canopy_trip_counts(), **{k: D.download(k) for k in ("t_veg", "cgrnd", "eflx_sh_veg", "h2ocan", "t_ref2m", "nrad", "albd", "albi", "fabd", "fabi", "ftid", "ftii", "ftdd", "flx_absdv", "flx_absdn", "flx_absiv", "flx_absin", "albsnd", "albsni", "albgrd", "albgri", "albsod", "vcmaxcintsun", "vcmaxcintsha", "fabd_sun_z", "fabi_sha_z", "fsun_z", "err_flags", "sabg_lyr", "sabv")})
