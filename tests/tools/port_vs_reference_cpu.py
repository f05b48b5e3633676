#!/usr/bin/env python3
"""CPU only: time the oracle (the port bench.py reports as cpu_baseline) against the reference's own headers
(oracle/_ref, built from /root/reference) on one thread, for the wrappers the reference builds here without netcdf.
python tests/tools/port_vs_reference_cpu.py [columns]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("OMP_NUM_THREADS", "1")
from elmkernels_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
ft = H.field_table_from_oracle()
cols, scal, soil = synth.make_state(ft, n, tier="B", seed=7)
R = O.Reference()
rows = [("frac_wet", lambda S: S.frac_wet(), lambda S: R.frac_wet(S)),
        ("canopy_hydrology", lambda S: S.canopy_hydrology(1800.0), lambda S: R.canopy_hydrology(S, 1800.0)),
        ("surface_radiation", lambda S: S.surface_radiation(), lambda S: R.surface_radiation(S)),
        ("canopy_temperature", lambda S: S.canopy_temperature(), lambda S: R.canopy_temperature(S)),
        ("bareground_fluxes", lambda S: S.bareground_fluxes(), lambda S: R.bareground_fluxes(S))]
print(f"{n} tier-B columns, 1 thread; ns per column")
for name, port, ref in rows:
    t = []
    for fn in (port, ref):
        best = 1e9
        for _ in range(3):
            S = H.oracle_state(cols, scal, soil)
            S.frac_wet()
            S.albedo_snicar()  # earlier wrappers of the step, so that inputs are realistic
            t0 = time.perf_counter()
            fn(S)
            best = min(best, time.perf_counter() - t0)
        t.append(best / n * 1e9)
    print(f"{name:20s} port {t[0]:8.1f}  reference {t[1]:8.1f}  port/reference {t[0] / t[1]:.2f}")
