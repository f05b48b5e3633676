#!/usr/bin/env python3
"""GPU box, development: a long advance() chain (cold start or synthetic state) on the device against the oracle, every field
compared bit for bit every few steps.  python tests/tools/soak_chain.py [cols] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from elmkernels_amd import state as st  # noqa: E402
from elmkernels_amd import synth  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
DT = 1800.0
ft = st.field_table()
cols, scal, soil = synth.make_state(ft, n, tier="B", seed=2026)
S = H.oracle_state(cols, scal, soil)
D = H.device_state(cols, scal, soil)
hgt = {k: S[k].copy() for k in ("forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch")}
rng = np.random.default_rng(1)
D.set_graph(True)
snl0 = S["snl"].copy()
for step in range(steps):
    e = rng.random(8)
    for k, v in hgt.items():
        D[k] = v
        S[k][...] = v
    st.get_forcing(D, 1.0 - e, e)
    S.get_forcing(1.0 - e, e, False)
    st.kokkos_init_timestep(D)
    S.init_timestep()
    st.advance_physics(D, DT)
    S.timestep7(DT)
    S.soil_temperature(DT)
    S.snow_hydrology(DT)
    S.surface_fluxes(DT)
    if step % 5 == 4 or step == steps - 1:
        worst, bad = H.compare_states(D, S, bitwise=True)
        flags, _ = D.error_summary()
        print(f"step {step + 1}: fields differing {len(bad)} {dict(list(bad.items())[:4])}; device flags {flags:#x} oracle flags {int(np.bitwise_or.reduce(S['err_flags'])):#x}; "
              f"snl changed in {int((S['snl'] != snl0).sum())} columns since the start", flush=True)
        if bad:
            sys.exit(1)
print("soak ok")
