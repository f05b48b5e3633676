#!/usr/bin/env python3
"""BASELINE.md section 4, step 1 - the reference CPU anchor (build container only: needs oracle/_ref, i.e. /root/reference).

The reference's OWN physics headers (oracle/_ref/libelmref.so + libelmref_canopy.so, g++ -O2 -fopenmp) behind
`#pragma omp parallel for` over columns - the execution shape of Kokkos::parallel_for(RangePolicy<OpenMP>(0, ncols))
(src/utils/invoke_kernel.hh:24-27, 40-46) - per wrapper and for the seven-wrapper sequence of ELMInterface::advance
(driver/kokkos/elm_kokkos_interface.cc:287-319), Tier-A synthetic columns, N = 100 000, 5 warm-up + 20 timed steps,
at 1 and 8 threads; beside each the oracle (the C restatement) on the same inputs.  Every timed call of a wrapper starts from
the state the wrappers before it left (the state is re-run through the whole sequence each step; t_veg and the forcing heights
are restored between steps as bench.py does).

    python tests/tools/reference_cpu_anchor.py [columns] [steps] > profiles/r04_reference_cpu_anchor.txt
"""
import os
import sys
import time

os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from elmkernels_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = 5
tier = os.environ.get("ANCHOR_TIER", "A")
ft = H.field_table_from_oracle()
cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=0x5EEDE1A0)
R = O.Reference()
DT = 1800.0
RESTORE = ["t_veg", "forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch"]

REF = [("frac_wet", lambda S: R.frac_wet(S)),
       ("albedo_snicar", lambda S: S.albedo_snicar_ref()),
       ("canopy_hydrology", lambda S: R.canopy_hydrology(S, DT)),
       ("surface_radiation", lambda S: R.surface_radiation(S)),
       ("canopy_temperature", lambda S: R.canopy_temperature(S)),
       ("bareground_fluxes", lambda S: R.bareground_fluxes(S)),
       ("canopy_fluxes", lambda S: S.canopy_fluxes_ref(DT))]
PORT = [("frac_wet", lambda S: S.frac_wet()),
        ("albedo_snicar", lambda S: S.albedo_snicar()),
        ("canopy_hydrology", lambda S: S.canopy_hydrology(DT)),
        ("surface_radiation", lambda S: S.surface_radiation()),
        ("canopy_temperature", lambda S: S.canopy_temperature()),
        ("bareground_fluxes", lambda S: S.bareground_fluxes()),
        ("canopy_fluxes", lambda S: S.canopy_fluxes(DT))]


def set_threads(t):
    O.lib().lib.elmo_set_threads(t)
    R.R.elmref_set_threads(t)
    O.lib().ref_canopy.elmref_canopy_set_threads(t)


def run(rows, threads):
    set_threads(threads)
    S = H.oracle_state(cols, scal, soil)
    saved = {k: S[k].copy() for k in RESTORE}
    acc = {name: 0.0 for name, _ in rows}
    total = 0.0
    for it in range(warm + steps):
        for k, v in saved.items():
            S[k][...] = v
        t_step = time.perf_counter()
        for name, fn in rows:
            t0 = time.perf_counter()
            fn(S)
            if it >= warm:
                acc[name] += time.perf_counter() - t0
        if it >= warm:
            total += time.perf_counter() - t_step
    return {k: v / steps / n * 1e9 for k, v in acc.items()}, total / steps / n * 1e9


print(f"# reference CPU anchor: {n} tier-{tier} columns, {warm} warm-up + {steps} timed steps, OMP_PROC_BIND=close OMP_WAIT_POLICY=passive;"
      f" host: {os.cpu_count()} hardware threads")
res = {}
for what, rows in (("reference", REF), ("port", PORT)):
    for t in (1, 8):
        res[(what, t)] = run(rows, t)
print("| kernel | reference 1 thread ns/col | reference 8 threads ns/col | reference 8-thread col-steps/s | port 1 thread ns/col | port 8 threads ns/col |")
print("|---|---|---|---|---|---|")
names = [r[0] for r in REF]
for name in names:
    r1, r8 = res[("reference", 1)][0][name], res[("reference", 8)][0][name]
    p1, p8 = res[("port", 1)][0][name], res[("port", 8)][0][name]
    print(f"| {name} | {r1:.0f} | {r8:.0f} | {1e9 / r8:.3e} | {p1:.0f} | {p8:.0f} |")
r1, r8 = res[("reference", 1)][1], res[("reference", 8)][1]
p1, p8 = res[("port", 1)][1], res[("port", 8)][1]
print(f"| **all 7** | **{r1:.0f}** | **{r8:.0f}** | **{1e9 / r8:.3e}** | {p1:.0f} | {p8:.0f} |")
print(f"# 8-thread parallel efficiency: reference {r1 / (8 * r8):.2f}, port {p1 / (8 * p8):.2f}; port/reference per core (1 thread): {p1 / r1:.2f}")
