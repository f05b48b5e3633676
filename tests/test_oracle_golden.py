"""Pin the oracle (oracle/*.c) against every golden vector the reference's own tests hold.

Mirrors test/test_{CanHydro,CanSunShade,SurfRad,CanTemp,BGFlux,CanFlux,SurfAlb}.cc: same inputs
(test/data/<Module>_IN.txt), same hard-wired LandType/dtime, same step ranges, same comparison
(IsAlmostEqual rel 1e-15 / abs 1e-20 against <Module>_OUT.txt) - one fixture step per column.

The reference's own CanopyFluxes run does not meet 1e-15 against its fixture (BASELINE.md section 2:
73 of 8633 comparisons differ, worst 3.5e-10 on h2ocan, steps {0,15,25,38,47,49}); the oracle is held to
the same picture: everything 1e-15-equal except a bounded set within 1e-9.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import fixtures as F


def _prepare(module):
    d = F.load(module)
    rows = F.select_steps(d, module)
    S = O.OracleState(len(rows))
    S.load_params()
    S.set_scalars(**F.TEST_LAND)
    fin, oin = F.split(d, "in/", rows, S.nlev)
    fout, _ = F.split(d, "out/", rows, S.nlev)
    F.fill_state(S, fin)
    S["vtype"][:] = F.TEST_LAND["vtype"]
    S["veg_active"][:] = 1
    return d, rows, S, oin, fout


def _compare(S, fout, n, rel=1e-15, skip=()):
    bad = {}
    total = 0
    for name, exp in fout.items():
        if name in skip:
            continue
        got = S[name].reshape(n, -1).astype(np.float64)
        ok = F.almost_equal(got, exp, rel) | np.isnan(exp)
        total += ok.size
        if not ok.all():
            bad[name] = (int((~ok).sum()), float(np.max(np.where(ok, 0.0, F.rel_err(got, exp)))))
    return total, bad


def test_canopy_hydrology_fixture():
    d, rows, S, oin, fout = _prepare("CanopyHydrology")
    S.set_scalars(oldfflag=int(oin["oldfflag"][0, 0]), dewmx=float(oin["dewmx"][0, 0]))
    assert (oin["oldfflag"] == oin["oldfflag"][0, 0]).all() and (oin["dewmx"] == oin["dewmx"][0, 0]).all()
    # test_CanHydro.cc:203-218 calls fraction_wet between ground_flux and snow_init; it only reads h2ocan,
    # which neither snow_init nor fraction_h2osfc touches, so wrapper order gives the same numbers
    S.canopy_hydrology(F.TEST_DTIME)
    S.frac_wet()
    total, bad = _compare(S, fout, len(rows))
    assert total >= 1824 and not bad, bad
    assert not S["err_flags"].any()


def test_canopy_sunshade_fixture():
    d, rows, S, oin, fout = _prepare("CanopySunShadeFractions")
    S.surface_radiation()
    total, bad = _compare(S, fout, len(rows))
    assert total >= 768 and not bad, bad


def test_surface_radiation_fixture():
    d, rows, S, oin, fout = _prepare("SurfaceRadiation")
    S.surface_radiation()
    total, bad = _compare(S, fout, len(rows))
    assert total >= 1440 and not bad, bad
    assert not S["err_flags"].any()


def test_canopy_temperature_fixture():
    d, rows, S, oin, fout = _prepare("CanopyTemperature")
    S.z0mr[:17] = oin["z0mr"][0]
    S.displar[:17] = oin["displar"][0]
    # the fixture tables are the first 17 PFTs of clm_params (numpft = 17): pins the .nc conversion too
    pft = np.load(F.GOLDEN + "/pft_params.npz")
    assert np.array_equal(oin["z0mr"][0], pft["z0mr"][:17]) and np.array_equal(oin["displar"][0], pft["displar"][:17])
    # test_CanTemp.cc:275-279: forcing heights are reset to the atmospheric height before the += in forcing_height
    for k in "utq":
        S[f"forc_hgt_{k}_patch"][:] = oin[f"forc_hgt_{k}"][:, 0]
    S.canopy_temperature()
    total, bad = _compare(S, fout, len(rows))
    assert total >= 2784 and not bad, bad


def test_bareground_fluxes_fixture():
    d, rows, S, oin, fout = _prepare("BareGroundFluxes")
    S["frac_veg_nosno"][:] = 0  # test_BGFlux.cc:219 "hardwire to make it run"
    S.bareground_fluxes_given(oin["forc_rho"][:, 0])
    total, bad = _compare(S, fout, len(rows), skip=("frac_veg_nosno",))
    assert total >= 2064 and not bad, bad


def test_bareground_fluxes_derived_rho_close():
    """The wrapper derives forc_rho (bareground_fluxes_kokkos.cc:31); the fixture carries ELM's own value."""
    d, rows, S, oin, fout = _prepare("BareGroundFluxes")
    import ctypes

    rho = O.lib().lib.elmo_derive_forc_rho
    rho.restype = ctypes.c_double
    rho.argtypes = [ctypes.c_double] * 3
    # forc_t is not in this fixture; thm = forc_t + 0.0098*forc_hgt_t_patch (canopy_temperature_impl.hh:295)
    forc_t = S["thm"] - 0.0098 * S["forc_hgt_t_patch"]
    got = np.array([rho(p, q, t) for p, q, t in zip(S["forc_pbot"], S["forc_qbot"], forc_t)])
    assert np.max(F.rel_err(got, oin["forc_rho"][:, 0])) < 1e-12


def test_surface_albedo_fixture():
    d, rows, S, oin, fout = _prepare("SurfaceAlbedo")
    assert (oin["albsat"] == oin["albsat"][0]).all() and (oin["albdry"] == oin["albdry"][0]).all()
    S.albsat[:] = oin["albsat"][0]
    S.albdry[:] = oin["albdry"][0]
    S["isoicol"][:] = 3
    # PFTDataAlb of vtype 12 from the converted .nc must equal the fixture's own rhol/rhos/taul/taus/xl rows
    v = F.TEST_LAND["vtype"]
    fx = np.r_[oin["rhol"][0].reshape(2, 17)[:, v], oin["rhos"][0].reshape(2, 17)[:, v],
               oin["taul"][0].reshape(2, 17)[:, v], oin["taus"][0].reshape(2, 17)[:, v], oin["xl"][0][v]]
    if not np.array_equal(fx, S.pft_alb[v]):
        fx = np.r_[oin["rhol"][0].reshape(17, 2)[v], oin["rhos"][0].reshape(17, 2)[v],
                   oin["taul"][0].reshape(17, 2)[v], oin["taus"][0].reshape(17, 2)[v], oin["xl"][0][v]]
    assert np.array_equal(fx, S.pft_alb[v])
    sun, sha = S.albedo_snicar_ex()
    total, bad = _compare(S, fout, len(rows), skip=("fabd_sun", "fabd_sha"))
    for name, got in (("fabd_sun", sun), ("fabd_sha", sha)):
        ok = F.almost_equal(got, fout[name]) | np.isnan(fout[name])
        total += ok.size
        assert ok.all(), name
    assert total >= 2350 and not bad, bad
    assert not S["err_flags"].any()
    sunlit = int((S["coszen"] > 0).sum())
    assert 0 < sunlit < len(rows)  # both day and night steps are exercised


def _run_canflux(given):
    d = F.load("CanopyFluxes")
    rows = F.select_steps(d, "CanopyFluxes")
    probe = O.OracleState(1)
    fin, oin = F.split(d, "in/", rows, probe.nlev)
    fout, _ = F.split(d, "out/", rows, probe.nlev)
    got = {k: np.zeros_like(v) for k, v in fout.items()}
    niters = []
    flags = 0
    # dayl / max_dayl are state scalars (elm_state.h:222) but vary per fixture step: one 1-column run per step
    for i in range(len(rows)):
        S = O.OracleState(1)
        S.load_params()
        S.set_scalars(**F.TEST_LAND, dayl=float(oin["dayl"][i, 0]), max_dayl=float(oin["max_dayl"][i, 0]))
        F.fill_state(S, {k: v[i : i + 1] for k, v in fin.items()})
        S["vtype"][:] = F.TEST_LAND["vtype"]
        if given:
            nit = S.canopy_fluxes_given(F.TEST_DTIME, oin["forc_rho"][i], oin["forc_po2"][i], oin["forc_pco2"][i], True)
        else:
            # forc_rho derived as the wrapper does; CO2/O2 partial pressures still from the fixture (see below)
            nit = S.canopy_fluxes_given(F.TEST_DTIME, None, oin["forc_po2"][i], oin["forc_pco2"][i], True)
        niters.append(int(nit[0]))
        flags |= int(S["err_flags"][0])
        for k in got:
            got[k][i] = S[k].reshape(1, -1)
    return d["steps"][rows], got, fout, np.array(niters), flags


def test_canopy_fluxes_fixture():
    steps, got, fout, niters, flags = _run_canflux(given=True)
    total = 0
    loose = {}
    for name, exp in fout.items():
        ok = F.almost_equal(got[name], exp) | np.isnan(exp)
        total += ok.size
        if not ok.all():
            r = np.where(ok, 0.0, F.rel_err(got[name], exp, floor=1e-18))
            loose[name] = (int((~ok).sum()), float(r.max()), sorted(set(steps[np.nonzero((~ok).any(axis=1))[0]].tolist())))
    nloose = sum(v[0] for v in loose.values())
    worst = max([v[1] for v in loose.values()] or [0.0])
    # reference vs the same fixture: 73 of 8633 beyond 1e-15, worst 3.5e-10
    assert total >= 8633
    assert nloose <= 120, loose
    assert worst < 1e-9, loose
    assert (flags & 0x7FF) == 0  # no throw/assert site of the reference is reached (Ball-Berry warning bit excluded)
    assert niters.min() >= 3 and niters.max() <= 41
    # both day (photosynthesis root-find active) and night steps occur
    par = F.load("CanopyFluxes")["in/parsun_z"]
    assert (par > 0).any() and (par <= 0).any()


def test_canopy_fluxes_derived_rho_close():
    """Deriving forc_rho as the L3 wrapper does (canopy_fluxes_kokkos.cc:49) instead of taking ELM's value
    moves results only at rounding level.  (forc_pco2 is NOT derivable from the fixture: ELM ran with
    397.84 ppm CO2 while the reference hard-wires CO2_PPMV = 355, elm_constants.h:31 - so the fixture's
    own partial pressures are kept.)"""
    d = F.load("CanopyFluxes")
    assert abs(float((d["in/forc_pco2"] / d["in/forc_pbot"]).mean()) * 1e6 - 397.84) < 0.01
    _, got_g, fout, nit_g, _ = _run_canflux(given=True)
    _, got_d, _, nit_d, _ = _run_canflux(given=False)
    assert np.abs(nit_g - nit_d).max() <= 1  # a last-bit change can move the convergence test by one iteration
    for name in fout:
        assert np.max(F.rel_err(got_g[name], got_d[name], floor=1e-12)) < 1e-6, name


def test_oracle_matches_the_committed_reference_outputs():
    """tests/golden/ref_branch_mix.npz: what the REFERENCE ITSELF (oracle/_ref, recorded in the build container by
    tests/refgolden.py) returns on 1 024 branch-mix columns for every stage of advance() its headers can run.  The oracle walks
    the same sequence here - also where /root/reference does not exist - and must reproduce every recorded field bit for bit."""
    from tests import _parity_mode
    from tests import refgolden as G

    fx = G.load()
    _, S = G.start_state()
    checked = 0
    for stage in G.STAGES:
        if str(fx[f"hash/{stage}"]) != G.state_hash(S):
            assert not _parity_mode.BITWISE_VALID, f"{stage}: the oracle chain left the recorded one on a host with the recorded libm"
            pytest.skip("another host libm: the oracle chain is not the one the fixture was recorded on")
        before = S.clone() if stage == "albedo_snicar" else None
        G.run_oracle(S, stage)
        keys = [k for k in fx.files if k.startswith(f"out/{stage}/")]
        for k in keys:
            got, exp = S[k.split("/")[2]], fx[k]
            assert ((got == exp) | (np.isnan(got.astype(float)) & np.isnan(exp.astype(float)))).all(), k
            checked += 1
        if stage == "albedo_snicar":
            day = before["coszen"] > 0
            assert np.array_equal(S["albsnd"][day], fx["out/snicar/albsnd"][day]) and np.array_equal(S["albsni"][day], fx["out/snicar/albsni"][day])
            for name, f, band, alb in (("flx_absdv", "flx_absd_snw", 0, "albsnd"), ("flx_absdn", "flx_absd_snw", 1, "albsnd"),
                                       ("flx_absiv", "flx_absi_snw", 0, "albsni"), ("flx_absin", "flx_absi_snw", 1, "albsni")):
                exp = fx[f"out/snicar/{f}"][:, :, band] * (1.0 - fx[f"out/snicar/{alb}"][:, band : band + 1])
                assert np.array_equal(S[name][day], exp[day]), name
            checked += 6
    assert str(fx["hash/evaluate_conservation"]) == G.state_hash(S)
    assert np.array_equal(S.evaluate_conservation(G.DT), fx["out/evaluate_conservation/diag"], equal_nan=True)
    assert checked >= 130
