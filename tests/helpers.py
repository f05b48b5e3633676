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
    S.snowage[...] = synth.snow_age_tables()
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
    D.set_snow_age_tables(synth.snow_age_tables())
    for k, v in cols.items():
        D[k] = v
    return D


# Outputs that are formed as a small difference of large operands carry the rounding error of the OPERANDS:
# the attainable agreement is 1e-12 of the operand magnitude, not of the (much smaller) result.  scale = that
# operand magnitude; the absolute floor used for the field is REL_TOL * scale.
CANCEL_SCALE = {
    # SNICAR layer absorption = difference of interface net fluxes, fractions of a unit incident flux
    "flx_absdv": 1e-2, "flx_absdn": 1e-2, "flx_absiv": 1e-2, "flx_absin": 1e-2,
    # 0.5 * (1 + erf(x)) with erf(x) ~ -1
    "frac_h2osfc": 1e-3,
    # canopy energy balance: rho*cp*conductance*(T_a - T_b) with |T| ~ 3e2 K  -> operands ~ 1e4 W/m2
    "eflx_sh_veg": 1e4, "eflx_sh_grnd": 1e4, "eflx_sh_snow": 1e4, "eflx_sh_soil": 1e4, "eflx_sh_h2osfc": 1e4,
    "eflx_sh_tot": 1e4, "dlrad": 1e3, "ulrad": 1e3,
    # vapour fluxes: rho*conductance*(q_a - q_b), operands ~ 1e-3 kg/m2/s;  h2ocan += dtime * (tran - evap)
    "qflx_tran_veg": 1e-3, "qflx_evap_veg": 1e-3, "qflx_evap_soi": 1e-3, "qflx_ev_snow": 1e-3, "qflx_ev_soil": 1e-3,
    "qflx_ev_h2osfc": 1e-3, "qflx_evap_tot": 1e-3, "h2ocan": 2.0,
}
# Bit-exactness.  Every <cmath> function the hot path calls - exp, log, log10, pow, atan, tanh, cos, erf, acos - is on the
# device a restatement of the host libm's algorithm (elmkernels_amd/csrc/elmk_math.h; sqrt and division are correctly
# rounded on both sides), so on bit-identical inputs every fp64 output must be BIT-IDENTICAL to the oracle's.  The set
# below lists outputs exempt from that (held to REL_TOL instead): none.
LIBM_RESIDUAL_FIELDS = set()


# soil_temperature phase change (used by its test only): what is left of a layer's ice, of a thin snow cover or of a
# pond after melting / freezing most of it - ice = max(0, ice - xm), h2osno = max(0, h2osno - xm),
# snow_depth *= h2osno_new / h2osno_old, h2osfc += xm - and latent-heat fluxes formed from such mass differences,
# hfus * (ice_before - ice_after) / dtime.  scale = magnitude of the operands (kg/m2, W/m2, kg/m2/s).
PHASE_CHANGE_SCALE = {
    "h2osoi_ice": 1e2, "h2osoi_liq": 1e2, "h2osno": 10.0, "snow_depth": 1.0, "h2osfc": 10.0, "int_snow": 10.0,
    "xmf": 1e3, "xmf_h2osfc": 1e3, "eflx_snomelt": 1e3, "eflx_h2osfc_snow": 1e3, "qflx_snomelt": 1e-2,
    "qflx_h2osfc_ice": 1e-2, "qflx_snow_melt": 1e-2, "qflx_snofrz": 1e-2, "qflx_snofrz_lyr": 1e-2,
}


def field_floor(name, extra_scale=None):
    scale = CANCEL_SCALE.get(name, 0.0)
    if extra_scale:
        scale = max(scale, extra_scale.get(name, 0.0))
    return max(ABS_FLOOR, REL_TOL * scale)


def compare_states(D, S, names=None, rel=REL_TOL, skip_cols=None, int_exact=True, extra_scale=None, bitwise=False):
    """Compare a device ELMState (download) with an OracleState field by field.

    -> (worst relative error, {field: (count over tolerance, worst rel err)}).  Integer fields must be equal.
    Tolerance per value: |a-b| <= rel*max(|a|,|b|) + field_floor(name).  bitwise=True: every fp64 field outside
    LIBM_RESIDUAL_FIELDS must match bit for bit (any NaN equals any NaN); the residual fields keep the tolerance.
    Columns in skip_cols (bool mask) are ignored (e.g. columns where either side raised a fatal flag)."""
    from tests import _parity_mode

    if bitwise and not _parity_mode.BITWISE_VALID:
        bitwise = False  # another host libm: the oracle is not the bits of a glibc-2.35 reference run (tests/_parity_mode.py)
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
        if bitwise and name not in LIBM_RESIDUAL_FIELDS:
            same = (got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))
            if not same.all():
                r = F.rel_err(got, exp, floor=0.0)
                worst = max(worst, float(r.max()))
                bad[name] = (int((~same).sum()), float(r.max()))
            continue
        r = F.rel_err(got, exp, floor=field_floor(name, extra_scale))
        m = float(r.max()) if r.size else 0.0
        worst = max(worst, m)
        if m > rel:
            bad[name] = (int((r > rel).sum()), m)
    return worst, bad
