"""Structural checks of the soil / snow temperature restatement (oracle/elmo_physics_d.c, CPU only).

The reference has no fixture for this path; the whole wrapper is pinned bit for bit by the reference's own per-column
functions in tests/test_oracle_vs_ref.py (test_soil_temperature_whole_wrapper_bitwise).  These are the reference-free
checks beside it:
  * the solver's output satisfies the assembled system;
  * the assembled system is consistent: with a uniform temperature profile and zero net surface heat flux the new
    temperatures equal the old ones (every row of the Crank-Nicolson matrix then reproduces T0 on its right-hand
    side; a wrong sign, index or snow / surface-water weighting breaks it);
  * one-sided forcing moves temperatures the right way and never by more than the forcing allows."""
import numpy as np

from elmkernels_amd import synth
from tests import helpers as H

DT = 1800.0


def _state(n=4000, seed=5):
    ft = H.field_table_from_oracle()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=seed)
    S = H.oracle_state(cols, scal, soil)
    S.timestep7(DT)
    return S


def _quiet(S, T0):
    """A copy of S with uniform temperature T0 and every surface heat flux term cancelled."""
    U = S.clone()
    U["t_soisno"][...] = np.where(np.arange(20)[None, :] >= 5 - S["snl"][:, None], T0, 0.0)
    U["t_h2osfc"][...] = T0
    for k in ("sabg_soil", "sabg_snow", "eflx_sh_soil", "eflx_sh_snow", "eflx_sh_h2osfc", "eflx_sh_grnd", "qflx_ev_soil",
              "qflx_ev_snow", "qflx_ev_h2osfc", "qflx_evap_soi"):
        U[k][...] = 0
    U["sabg_lyr"][...] = 0
    U["frac_veg_nosno"][...] = 1  # no direct atmospheric longwave on the ground
    U["dlrad"][...] = U["emg"] * 5.67e-8 * T0 ** 4  # balances the emitted longwave
    return U


def test_solution_satisfies_the_assembled_system():
    S = _state()
    e = S.soil_temperature_ex(DT)
    act = np.arange(21)[None, :] >= 5 - e_snl(S)[:, None]
    res = np.zeros_like(e["rhs"])
    for i in range(21):
        for band, off in ((0, 2), (1, 1), (2, 0), (3, -1), (4, -2)):
            j = i + off
            if 0 <= j < 21:
                res[:, i] += e["lhs"][:, i, band] * e["sol"][:, j]
    rel = np.abs(res - e["rhs"]) / np.maximum(1.0, np.abs(e["rhs"]))
    assert rel[act].max() < 1e-12
    assert (e["sol"][~act] == 0).all()  # rows above the snow pack stay zero (A, B, Z are zero-filled)


def e_snl(S):
    return S["snl"]


def test_uniform_profile_without_forcing_is_a_fixed_point():
    S = _state()
    before = S["snl"].copy()
    for T0 in (268.0, 271.3, 285.0):
        U = _quiet(S, T0)
        e = U.soil_temperature_ex(DT)
        assert np.abs(e["hs"][:, :3]).max() < 1e-9
        act = np.arange(21)[None, :] >= 5 - before[:, None]
        err = np.abs(np.where(act, e["sol"] - T0, 0.0)).max(axis=1)
        dry = S["frac_h2osfc"] == 0
        assert dry.sum() > 1000 and err[dry].max() < 1e-11
        # With standing surface water the reference's top-soil row is not balanced (get_matrix_soil adds a
        # frac_h2osfc term to the diagonal that get_rhs_soil has no counterpart for, soil_temp_lhs_impl.hh:285-290
        # vs soil_temp_rhs_impl.hh:135-177); the restatement reproduces that, so those columns are not a fixed point.
        assert err[~dry].max() > 1e-3


def test_forcing_moves_temperature_the_right_way():
    S = _state()
    U = _quiet(S, 270.0)
    warm = U.clone()
    warm["sabg_soil"][...] = 100.0
    warm["sabg_snow"][...] = 100.0
    warm["sabg_lyr"][...] = 0.0
    top = 5 - S["snl"]
    warm["sabg_lyr"][np.arange(S.ncols), top] = 100.0
    e0 = U.soil_temperature_ex(DT)
    e1 = warm.soil_temperature_ex(DT)
    dry = (S["frac_h2osfc"] == 0)
    row = np.where(S["snl"] > 0, top, 6)  # first active snow row, or the first soil row (row 5 is surface water)
    d = (e1["sol"] - e0["sol"])[np.arange(S.ncols), row]
    assert (d[dry] > 0).all() and d[dry].max() < 100.0
    deep = (e1["sol"] - e0["sol"])[:, 20]
    assert (np.abs(deep[dry]) < np.abs(d[dry])).all()  # the signal decays with depth
