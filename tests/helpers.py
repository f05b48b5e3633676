"""Shared plumbing for the parity tests: build identical states in the oracle and in libelmk, compare fields."""
import numpy as np

from elmkernels_amd import state as st
from elmkernels_amd import synth
from oracle import oracle as O
from tests import fixtures as F

# fp64 parity bar of BASELINE.json / SURVEY.md section 8(c)
REL_TOL = 1e-12
ABS_FLOOR = 1e-18


def field_table_from_oracle():
    """{name: (id, nlev, dtype)} derived from the oracle's registry (usable without the HIP library)."""
    S = O.OracleState(1)
    return {k: (i, S.nlev[k], S.fields[k].dtype.type) for i, k in enumerate(S.fields) if k != "err_flags"}


def oracle_state(cols, scal, soil, land=None, pft=None, optics=None):
    n = next(iter(cols.values())).shape[0]
    S = O.OracleState(n)
    S.load_params(pft, optics)
    S.albsat[:] = soil["albsat"]
    S.albdry[:] = soil["albdry"]
    S.set_scalars(**(land or synth.TEST_LAND), **scal)
    for k, v in cols.items():
        S.fields[k][...] = v
    return S


def device_state(cols, scal, soil, land=None, device=0):
    n = next(iter(cols.values())).shape[0]
    D = st.ELMState(n, device)
    pft, optics = synth.load_params()
    D.set_pft(pft)
    D.set_snicar(optics)
    D.set_soilcolor(soil["albsat"], soil["albdry"])
    D.set_land(**(land or synth.TEST_LAND))
    D.set_scalars(**scal)
    for k, v in cols.items():
        D[k] = v
    return D


def compare_states(D, S, names=None, rel=REL_TOL, floor=ABS_FLOOR, skip_cols=None, int_exact=True):
    """Compare a device ELMState (download) with an OracleState field by field.

    -> (worst relative error, {field: (count over tolerance, worst rel err)}).  Integer fields must be equal.
    Columns in skip_cols (bool mask) are ignored (e.g. columns where either side raised a fatal flag)."""
    worst = 0.0
    bad = {}
    for name in names or [k for k in S.fields if k != "err_flags"]:
        got = D[name]
        exp = S.fields[name]
        if skip_cols is not None:
            got = got[~skip_cols]
            exp = exp[~skip_cols]
        if exp.dtype.kind in "iu":
            if int_exact and not np.array_equal(got, exp):
                bad[name] = (int((got != exp).sum()), float("inf"))
            continue
        r = F.rel_err(got, exp, floor=floor)
        m = float(r.max()) if r.size else 0.0
        worst = max(worst, m)
        if m > rel:
            bad[name] = (int((r > rel).sum()), m)
    return worst, bad
