#!/usr/bin/env python3
"""BASELINE config 5's second half, report only: what an fp32 STATE would do to the results.  The kernels keep computing in
fp64; every fp64 state field is rounded to the nearest fp32 value before each step (what storing the state in fp32 and
widening on load would give), the seven wrappers run, and the outputs are compared with the all-fp64 chain after 1 and after
10 steps.  Also the HBM bytes an fp32 state would move (from the PMC traffic tables of the fp64 build: every fp64 state byte
halves, the canopy_fluxes queue records and int fields stay).  python tests/tools/fp32_state_study.py [cols] [tier]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
tier = sys.argv[2] if len(sys.argv) > 2 else "B"
D64, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
D32, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
f64_fields = [k for k, (fid, nlev, dt) in D64.fields.items() if dt == np.float64]
OUT = ["t_veg", "t_grnd", "h2ocan", "eflx_sh_tot", "eflx_lh_tot", "qflx_evap_tot", "fsa", "sabv", "albd", "btran", "t_ref2m", "q_ref2m",
       "cgrnd", "dlrad", "ulrad", "h2osno", "snow_depth", "frac_sno", "qflx_tran_veg", "eflx_sh_veg"]


def round_state(D):
    for k in f64_fields:
        a = D.download(k)
        with np.errstate(over="ignore"):
            r = a.astype(np.float32).astype(np.float64)
        r = np.where(np.isfinite(r) | ~np.isfinite(a), r, a)  # (1e36 "special values" overflow fp32: keep them)
        D.upload(k, r)


def report(step):
    print(f"after {step} step(s): relative difference |fp32-state - fp64| / max(|fp64|, floor) per output field")
    worst = 0.0
    for k in OUT:
        a, b = D64.download(k).astype(np.float64), D32.download(k).astype(np.float64)
        floor = max(1e-12, 1e-6 * float(np.nanmax(np.abs(a)))) if a.size else 1.0
        rel = np.abs(a - b) / np.maximum(np.abs(a), floor)
        rel = rel[np.isfinite(rel)]
        print(f"  {k:16s} median {np.median(rel):.2e}  99% {np.percentile(rel, 99):.2e}  max {rel.max():.2e}")
        worst = max(worst, float(np.percentile(rel, 99)))
    trips64, trips32 = D64.canopy_trip_counts(), D32.canopy_trip_counts()
    print(f"  leaf-temperature trip counts differ in {int((trips64 != trips32).sum())} of {cols} columns; worst 99th-percentile difference {worst:.2e}")


for step in range(1, 11):
    round_state(D32)
    st.timestep7_fused(D32, 1800.0)
    st.timestep7_fused(D64, 1800.0)
    if step in (1, 10):
        report(step)

# traffic an fp32 state would move, from the PMC tables of the fp64 build
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles")
try:
    t = json.load(open(os.path.join(root, f"{bench.PROFILE_TAG}_hbm_traffic_pmc_fused_tier{tier}.json")))["kernels"]
    tot = sum(v["hbm_bytes_per_launch"] for k, v in t.items() if "k_copy" not in k) / 1e6
    rec = (46 + 24) * 8 * 2 + 3 * 4 * 2  # canopy_fluxes queue records (written once, read once) + int records: not state
    ints = 60  # int / bool state traffic per column (snl, nrad, frac_veg_nosno, vtype ... read or written by the step)
    print(f"HBM traffic of the fused fp64 step (PMC, 1 M columns tier {tier}): {tot:.0f} B/column; with an fp32 state about "
          f"{rec + ints + (tot - rec - ints) / 2:.0f} B/column (the {rec} B of fp64 queue records and ~{ints} B of integer fields stay)")
except OSError:
    pass
