#!/usr/bin/env python3
"""GPU box: the PCIe-inclusive rate of one model step as a host driver would run it through the C ABI - upload of the
14 forcing fields the reference's init_timestep rewrites every step, the seven wrappers, download of the 19 PrimaryVars
fields (src/data/elm_state.h:17-48) - next to the HBM-resident rate bench.py reports.  Host buffers are pageable numpy
arrays in the reference's [column][level] layout, i.e. what elmk_upload / elmk_download take from a C++ caller.
python tests/tools/pcie_rate.py [cols] [steps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
FORCING = ["forc_hgt_q_patch", "forc_hgt_t_patch", "forc_hgt_u_patch", "forc_lwrad", "forc_pbot", "forc_qbot", "forc_rain",
           "forc_snow", "forc_solad", "forc_solai", "forc_tbot", "forc_thbot", "forc_u", "forc_v"]
PRIMARY = ["snl", "snow_depth", "frac_sno", "int_snow", "snw_rds", "h2osoi_liq", "h2osoi_ice", "h2osoi_vol", "h2ocan", "h2osno",
           "h2osfc", "t_soisno", "t_grnd", "t_h2osfc", "t_h2osfc_bef", "nrad", "dz", "zsoi", "zisoi"]
D, _ = bench.build_state(cols, 0, "A", 0x5EEDE1A0)
host = {k: D.download(k) for k in FORCING}
out = {k: D.download(k) for k in PRIMARY}
up_bytes = sum(v.nbytes for v in host.values())
down_bytes = sum(v.nbytes for v in out.values())


def step(transfers):
    D.restore_fields()
    if transfers:
        for k, v in host.items():
            D.upload(k, v)
    st.timestep7(D, 1800.0)
    if transfers:
        for k in PRIMARY:
            D.download(k, out=out[k])


res = {}
for name, tr in (("resident", False), ("pcie_inclusive", True)):
    for _ in range(4):  # warm-up: scheduling hints of canopy_fluxes, staging buffers
        step(tr)
    D.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(tr)
    D.sync()
    res[name] = (time.perf_counter() - t0) / steps
print(json.dumps({
    "columns": cols, "steps": steps, "upload_bytes_per_column": up_bytes / cols, "download_bytes_per_column": down_bytes / cols,
    "resident_ms_per_step": res["resident"] * 1e3, "pcie_inclusive_ms_per_step": res["pcie_inclusive"] * 1e3,
    "resident_gridcell_timesteps_per_s": cols / res["resident"], "pcie_inclusive_gridcell_timesteps_per_s": cols / res["pcie_inclusive"],
    "effective_host_transfer_GBps": (up_bytes + down_bytes) / max(res["pcie_inclusive"] - res["resident"], 1e-9) / 1e9,
    "note": "pageable host memory, one 32 MiB staging buffer, synchronous hipMemcpy per chunk + layout conversion kernel",
}))
D.close()
