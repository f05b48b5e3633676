#!/usr/bin/env python3
"""Convert the reference's own known-answer fixtures into compact .npz vectors.

Run in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py            # fixtures -> tests/golden/*.npz
    python tests/golden/make_golden.py --ref      # + branch-coverage vectors from oracle/_ref

What is converted (data only - inputs and expected outputs; no reference source):

  test/data/<Module>_IN.txt / _OUT.txt   ELM (Fortran) single-column dumps, block format
                                         "NSTEP n" ... "!!! n", one "label v0 v1 ..." line per
                                         variable (reference reader: src/utils/read_test_input.cc:14-22,
                                         src/utils/read_test_input.hh:42-68)
      -> <Module>.npz   keys "in/<label>" and "out/<label>", shape [nsteps, nvalues] (float64;
                        the int/bool labels hold integral values), plus "steps" (the NSTEP ids)
  test/data/SnowOptics_IN.txt            SNICAR Mie / BC lookup tables (same text format)
      -> SnowOptics.npz one key per table, flat float64
  test/data/clm_params_c180524.nc        NetCDF-3 classic PFT parameter file (read with
                                         scipy.io.netcdf_file, which executes nothing from the file)
      -> pft_params.npz the 40 per-PFT vectors the hot path uses (25 PFTs; tc_stress has extent 1)

With --ref, the compiled reference (oracle/_ref/libelmref.so, built from the reference headers where
they lie by oracle/Makefile) is run on seeded synthetic columns that reach the branches the bundled
fixtures never take (snow layers, bare ground, capped snow, ponded water ...) and inputs + outputs
are stored as ref_<kernel>.npz.  Those are outputs of the reference itself, generated here because
/root/reference cannot travel to the GPU box.
"""
import argparse
import os
import sys

import numpy as np

REF_DATA = "/root/reference/test/data"
HERE = os.path.dirname(os.path.abspath(__file__))

MODULES = [
    "CanopyHydrology",
    "CanopySunShadeFractions",
    "SurfaceRadiation",
    "CanopyTemperature",
    "BareGroundFluxes",
    "CanopyFluxes",
    "SurfaceAlbedo",
]

PFT_VARS = (
    "fnr act25 kcha koha cpha vcmaxha jmaxha tpuha lmrha vcmaxhd jmaxhd tpuhd lmrhd lmrse qe "
    "theta_cj bbbopt mbbopt c3psn slatop leafcn flnr fnitr dleaf smpso smpsc tc_stress z0mr displar "
    "xl roota_par rootb_par rholvis rholnir rhosvis rhosnir taulvis taulnir tausvis tausnir"
).split()


def parse_blocks(path):
    """Return (steps, {label: [row per step]}) for one NSTEP-block text fixture."""
    steps = []
    cols = {}
    cur = None
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "NSTEP":
                cur = int(tok[1])
                steps.append(cur)
                continue
            if tok[0] == "!!!":
                cur = None
                continue
            if cur is None:
                continue
            label = tok[0]
            vals = np.array([float(v) for v in tok[1:]], dtype=np.float64)
            rows = cols.setdefault(label, {})
            # first occurrence wins, as in the reference reader (it returns at the first label match)
            rows.setdefault(cur, vals)
    return steps, cols


def stack(steps, rows):
    n = max(len(v) for v in rows.values())
    out = np.full((len(steps), n), np.nan)
    for i, s in enumerate(steps):
        if s in rows and len(rows[s]) == n:
            out[i] = rows[s]
    return out


def convert_module(mod, src=REF_DATA, dst=HERE):
    data = {}
    steps_in, cin = parse_blocks(os.path.join(src, mod + "_IN.txt"))
    steps_out, cout = parse_blocks(os.path.join(src, mod + "_OUT.txt"))
    assert steps_in == steps_out, mod
    data["steps"] = np.array(steps_in, dtype=np.int64)
    for label, rows in cin.items():
        data["in/" + label] = stack(steps_in, rows)
    for label, rows in cout.items():
        data["out/" + label] = stack(steps_out, rows)
    os.makedirs(dst, exist_ok=True)
    np.savez_compressed(os.path.join(dst, mod + ".npz"), **data)
    print(f"{mod}: {len(steps_in)} steps, {len(cin)} in / {len(cout)} out labels -> {os.path.relpath(dst, HERE) or '.'}")


def convert_snowoptics():
    steps, cols = parse_blocks(os.path.join(REF_DATA, "SnowOptics_IN.txt"))
    data = {label: rows[steps[0]] for label, rows in cols.items() if label != "dtime"}
    np.savez_compressed(os.path.join(HERE, "SnowOptics.npz"), **data)
    print("SnowOptics:", {k: v.shape for k, v in list(data.items())[:4]}, "...")


def convert_pft():
    import scipy.io

    f = scipy.io.netcdf_file(os.path.join(REF_DATA, "clm_params_c180524.nc"), "r", mmap=False)
    data = {n: np.array(f.variables[n].data, dtype=np.float64).reshape(-1) for n in PFT_VARS}
    np.savez_compressed(os.path.join(HERE, "pft_params.npz"), **data)
    print("pft_params:", len(data), "vars")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", action="store_true", help="also generate branch vectors from oracle/_ref")
    args = ap.parse_args()
    if not os.path.isdir(REF_DATA):
        sys.exit("reference data not mounted; the committed .npz files are the fixtures")
    for m in MODULES:
        convert_module(m)
    # test/new_data: a second ELM dump of the same site under different (snow-free, summer) forcing that no reference test
    # reads; same format, converted the same way into tests/golden/newdata/
    for m in MODULES:
        convert_module(m, REF_DATA.replace("/data", "/new_data"), os.path.join(HERE, "newdata"))
    convert_snowoptics()
    convert_pft()
    if args.ref:
        sys.path.insert(0, os.path.join(HERE, "..", ".."))
        from tests import refgen  # noqa: E402

        refgen.generate(HERE)


if __name__ == "__main__":
    main()
