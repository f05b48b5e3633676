#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wrapper, per-field worst relative/absolute difference HIP vs oracle, no asserts.

    python tests/tools/parity_report.py [tier] [ncols] [seed]  > gpurun_out/parity_report.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from elmkernels_amd import state as st  # noqa: E402
from elmkernels_amd import synth  # noqa: E402
from tests import fixtures as F  # noqa: E402
from tests import helpers as H  # noqa: E402

tier = sys.argv[1] if len(sys.argv) > 1 else "B"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4
DT = 1800.0

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
]
print(f"tier {tier} n {n} seed {seed}")
for name, dev, ora in calls:
    dev()
    ora()
    fd, fo = D["err_flags"], S["err_flags"]
    print(f"== {name}: flags dev {np.unique(fd)} oracle {np.unique(fo)} mismatch {(fd != fo).sum()}")
    for k, exp in S.fields.items():
        if k == "err_flags":
            continue
        got = D[k]
        if exp.dtype.kind in "iu":
            if not np.array_equal(got, exp):
                print(f"   INT {k}: {(got != exp).sum()} differ")
            continue
        r = F.rel_err(got, exp, floor=0.0)
        if r.max() > 1e-14:
            i = np.unravel_index(np.argmax(r), r.shape)
            a = np.abs(got.astype(float) - exp)
            print(f"   {k:22s} worst rel {r.max():.2e} at {i} (dev {got[i]!r} ora {exp[i]!r}); n>1e-12: {(r > 1e-12).sum()}"
                  f"; worst abs {np.nanmax(a):.2e}")
    for k, v in S.fields.items():
        if k != "err_flags":
            D[k] = v
    D.clear_errors()
    S["err_flags"][:] = 0
D.close()
