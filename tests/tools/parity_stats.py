#!/usr/bin/env python3
"""GPU box: per-wrapper parity statistics against the oracle on re-synchronised inputs - for every wrapper, the number of
fp64 output values that are bit-identical, within 1e-12, and the worst relative error.
python tests/tools/parity_stats.py [n] [tier] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from elmkernels_amd import state as st  # noqa: E402
from elmkernels_amd import synth  # noqa: E402
from tests import fixtures as F  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tier = sys.argv[2] if len(sys.argv) > 2 else "B"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4
DT = synth.DTIME
DETAIL = os.environ.get("DETAIL", "0") != "0"
ft = st.field_table()
cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=seed)
S = H.oracle_state(cols, scal, soil)
D = H.device_state(cols, scal, soil)
calls = [
    ("frac_wet", lambda: st.kokkos_frac_wet(D), S.frac_wet),
    ("albedo_snicar", lambda: st.kokkos_albedo_snicar(D), S.albedo_snicar),
    ("canopy_hydrology", lambda: st.kokkos_canopy_hydrology(D, DT), lambda: S.canopy_hydrology(DT)),
    ("surface_radiation", lambda: st.kokkos_surface_radiation(D), S.surface_radiation),
    ("canopy_temperature", lambda: st.kokkos_canopy_temperature(D), S.canopy_temperature),
    ("bareground_fluxes", lambda: st.kokkos_bareground_fluxes(D), S.bareground_fluxes),
    ("canopy_fluxes", lambda: st.kokkos_canopy_fluxes(D, DT), lambda: S.canopy_fluxes(DT)),
    ("soil_temperature", lambda: st.kokkos_soil_temperature(D, DT), lambda: S.soil_temperature(DT)),
    ("surface_fluxes", lambda: st.kokkos_surface_fluxes(D, DT), lambda: S.surface_fluxes(DT)),
]
print(f"{n} columns, tier {tier}, seed {seed}")
for name, dev, ora in calls:
    before = {k: v.copy() for k, v in S.fields.items()}
    dev()
    ora()
    tot = nbit = n12 = 0
    detail = []
    worst, worst_name = 0.0, ""
    for k, v in S.fields.items():
        if k == "err_flags" or v.dtype != np.float64:
            continue
        changed = not np.array_equal(v, before[k], equal_nan=True)
        d = D[k]
        if not changed and np.array_equal(d, v, equal_nan=True):
            continue
        a, b = np.asarray(d).ravel(), np.asarray(v).ravel()
        same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        r = F.rel_err(a, b, floor=0.0)
        tot += a.size
        nbit += int(same.sum())
        n12 += int((r <= 1e-12).sum())
        if r.max() > worst:
            worst, worst_name = float(r.max()), k
        if DETAIL and not same.all():
            detail.append(f"{k}: {int((~same).sum())} ({r.max():.1e})")
    print(f"  {name:20s} values {tot:9d}  bit-identical {nbit / max(tot, 1) * 100:8.4f} %  within 1e-12 {n12 / max(tot, 1) * 100:8.4f} %  "
          f"worst rel {worst:.2e} ({worst_name})")
    if detail:
        print("      not bit-identical: " + ", ".join(detail))
    for k, v in S.fields.items():  # re-synchronise: the next wrapper starts from bit-identical inputs
        if k != "err_flags":
            D[k] = v
D.close()
