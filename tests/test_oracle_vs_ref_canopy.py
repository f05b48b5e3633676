"""The two dominant wrappers of the hot path - kokkos_albedo_snicar and kokkos_canopy_fluxes (photosynthesis inside) - run by
the REFERENCE'S OWN functions (oracle/_ref/libelmref_canopy.so = canopy_fluxes.h, photosynthesis.h, surface_albedo.h and
snow_snicar.h of /root/reference compiled by oracle/Makefile; oracle/ref_harness_canopy.cc says how they build without netcdf)
against the oracle restatement, on seeded synthetic columns that reach what the reference's single-site fixtures never take:
C4 plants, every plant type of the parameter file, soybean land units, Brent's method, the itmax fall-back, night, snow packs
of 0..5 layers, lake / land-ice / wetland / urban land units.

Same compiler family, same libm, same operation order => the bar is bit-for-bit equality of every state field.
Skipped (not failed) where the reference library was not built (it cannot be built on the GPU box; the prebuilt .so travels).
"""
import ctypes as C

import numpy as np
import pytest

from elmkernels_amd import synth
from oracle import oracle as O
from tests import fixtures as F
from tests import helpers as H

pytestmark = pytest.mark.skipif(not O.have_ref_canopy(), reason="oracle/_ref/libelmref_canopy.so not built here")

DT = 1800.0
REF_THREW = np.uint32(1 << 31)


def _diff(A, B, skip=("err_flags",)):
    out = {}
    for k in A.fields:
        if k in skip:
            continue
        a, b = A.fields[k], B.fields[k]
        eq = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
        if not eq.all():
            out[k] = (int((~eq).sum()), float(np.max(F.rel_err(a, b, floor=0.0))))
    return out


def _state(n, seed, land=None, every_pft=False):
    ft = H.field_table_from_oracle()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=seed)
    if every_pft:  # every row of the parameter file that has leaves (1..24), C3 and C4, trees to crops
        cols["vtype"] = np.random.default_rng(seed).integers(1, 25, n).astype(np.int32)
    return H.oracle_state(cols, scal, soil, land=land)


def _upto_canopy_fluxes(S):
    S.frac_wet()
    S.albedo_snicar()
    S.canopy_hydrology(DT)
    S.surface_radiation()
    S.canopy_temperature()
    S.bareground_fluxes()


LANDS = (
    dict(ltype=1, ctype=1, vtype=12, urbpoi=0, lakpoi=0),   # soil
    dict(ltype=2, ctype=0, vtype=15, urbpoi=0, lakpoi=0),   # crop
    dict(ltype=1, ctype=1, vtype=23, urbpoi=0, lakpoi=0),   # soybean (photosynthesis_impl.hh: the soybean bbbopt branch)
    dict(ltype=3, ctype=0, vtype=0, urbpoi=0, lakpoi=0),    # land ice
    dict(ltype=5, ctype=0, vtype=0, urbpoi=0, lakpoi=1),    # deep lake (soil_albedo's frozen / unfrozen lake forms)
    dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0),    # wetland
    dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0),   # urban: every routine of the albedo wrapper returns early
)


@pytest.mark.parametrize("land", LANDS, ids=lambda d: f"ltype{d['ltype']}_v{d['vtype']}")
def test_albedo_snicar_whole_wrapper_bitwise(land):
    """albedo_kokkos.cc:10-376 - init_timestep, soil_albedo, both SNICAR passes, ground_albedo, flux_absorption_factor,
    canopy_layer_lai, two_stream_solver - by the reference's functions against elmo_albedo_snicar: every state field and the
    wrapper-local fabd_sun / fabd_sha bit for bit."""
    A = _state(6016, 77 + land["ltype"], land, every_pft=True)
    A.frac_wet()
    B = A.clone()
    sun_a, sha_a = A.albedo_snicar_ex()
    sun_b, sha_b = B.albedo_snicar_ref()
    assert not (B["err_flags"] & REF_THREW).any()
    d = _diff(A, B)
    assert not d, d
    assert np.array_equal(sun_a.view(np.uint64), sun_b.view(np.uint64)) and np.array_equal(sha_a.view(np.uint64), sha_b.view(np.uint64))
    if not land["urbpoi"]:
        # the columns are a real mix: sunlit and dark, packs of every depth, snow-free ground
        assert (A["coszen"] > 0).mean() > 0.2 and (A["coszen"] <= 0).mean() > 0.2
        assert set(np.unique(A["snl"])) == {0, 1, 2, 3, 4, 5}
        assert (A["albd"] != 1.0).any() and (A["albgrd"] != A["albsod"]).any()  # canopy / snow really changed the surface


@pytest.mark.parametrize("land", LANDS[:4] + LANDS[5:6], ids=lambda d: f"ltype{d['ltype']}_v{d['vtype']}")
def test_canopy_fluxes_whole_wrapper_bitwise(land):
    """canopy_fluxes_kokkos.cc:6-265 - initialize_flux, stability_iteration (with photosynthesis, hybrid, brent, ci_func,
    quadratic), compute_flux - by the reference's functions against elmo_canopy_fluxes, after the six wrappers that precede it
    in the step, and chained into a second step: every state field bit for bit.  The inputs reach C4 photosynthesis, all 24
    leafed plant types, Brent's method and day and night columns (counted by the restatement's branch counters)."""
    A = _state(12032, 5 + land["ltype"] + land["vtype"], land, every_pft=True)
    counts = dict(hybrid=0, brent=0, itmax=0, c4=0)
    for step in range(2):
        _upto_canopy_fluxes(A)
        B = A.clone()
        O.psn_counters(reset=True)
        A.canopy_fluxes(DT)
        for k, v in O.psn_counters().items():
            counts[k] += v
        B.canopy_fluxes_ref(DT)
        threw = (B["err_flags"] & REF_THREW) != 0
        # the reference throws where the restatement raises a fatal photosynthesis flag, and nowhere else
        fatal = (A["err_flags"] & np.uint32(0b11100)) != 0
        assert np.array_equal(threw, fatal), (int(threw.sum()), int(fatal.sum()))
        assert threw.mean() < 0.01
        keep = ~threw
        d = {}
        for k in A.fields:
            if k == "err_flags":
                continue
            a, b = A.fields[k][keep], B.fields[k][keep]
            eq = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
            if not eq.all():
                d[k] = (int((~eq).sum()), float(np.max(F.rel_err(a, b, floor=0.0))))
        assert not d, (step, d)
        for k in A.fields:  # a column in which the reference threw was left half-written: continue from the restatement's state
            B.fields[k][...] = A.fields[k]
    veg = land["ltype"] in (1, 2)
    if veg:
        assert counts["hybrid"] > 20000 and counts["c4"] > 2000 and counts["brent"] > 20, counts


def _psn_inputs(rng, n, extreme):
    u = lambda lo, hi: lo + (hi - lo) * rng.random(n)
    if not extreme:  # wider than any model state, still physical
        t_veg = u(230.0, 330.0)
        pbot = u(5.0e4, 1.05e5)
        esat = 611.0 * np.exp(17.27 * (t_veg - 273.15) / (t_veg - 35.85))
        cols = [u(0.01, 8.0), np.where(rng.random(n) < 0.15, 0.0, 10.0 ** u(-2.0, 2.7)), u(0.01, 6.0),  # tlai_z, par_z (15 % dark), lai_z
                pbot, t_veg, u(240.0, 310.0),                                                            # forc_pbot, t_veg, t10
                esat, esat * u(0.02, 1.2),                                                               # esat_tv, eair (also super-saturated)
                0.209 * pbot, 10.0 ** u(-4.6, -3.0) * pbot,                                              # oair, cair (25..1000 ppm)
                10.0 ** u(0.0, 3.0), np.where(rng.random(n) < 0.1, 0.0, rng.random(n)),                  # rb, btran
                u(0.01, 1.0), t_veg + u(-15.0, 15.0), u(0.05, 3.0)]                                      # dayl_factor, thm, vcmaxcint
    else:  # far outside: the only inputs found that exhaust the secant iteration (about one call in a million)
        t_veg = u(200.0, 350.0)
        pbot = 10.0 ** u(3.5, 5.2)
        esat = 611.0 * np.exp(17.27 * (t_veg - 273.15) / (t_veg - 35.85))
        cols = [u(0.001, 12.0), 10.0 ** u(-4.0, 3.5), u(0.001, 10.0), pbot, t_veg, u(200.0, 330.0), esat, esat * 10.0 ** u(-3.0, 0.5),
                0.209 * pbot * 10.0 ** u(-2.0, 1.0), 10.0 ** u(-7.0, -1.0) * pbot, 10.0 ** u(-2.0, 5.0), 10.0 ** u(-6.0, 0.0),
                10.0 ** u(-3.0, 0.0), t_veg + u(-40.0, 40.0), 10.0 ** u(-3.0, 1.0)]
    return np.stack(cols, axis=1).copy()


def test_photosynthesis_alone_bitwise_over_wide_inputs():
    """photosynthesis() itself (photosynthesis_impl.hh:9-283), 1.4 million independent calls: 400 000 over ranges wider than a
    model state offers - leaf temperatures 230..330 K, light from darkness to full sun, closed to open stomata (btran 0..1),
    boundary-layer resistances over three decades, every plant type - and a million far outside them, which is what it takes
    to reach the itmax exit of hybrid() (:606-612).  ci_z and rs bit for bit; the reference throws nowhere and the restatement
    raises no flag.  Counted by the restatement: C4 calls, calls that end in Brent's method, calls that exhaust the iteration."""
    S = O.OracleState(1)
    S.load_params()
    L = O.lib()
    table = L.lib.elmo_pft_psn_ptr
    table.restype = C.c_void_p
    table.argtypes = [C.c_void_p]
    tp = table(S.ptr)
    L.lib.elmo_photosynthesis_batch.argtypes = [C.c_int64] + [C.c_void_p] * 6
    total = dict(hybrid=0, brent=0, itmax=0, c4=0)
    for seed, n, extreme in ((99, 400_000, False), (1004, 500_000, True), (1005, 500_000, True)):
        rng = np.random.default_rng(seed)
        x = _psn_inputs(rng, n, extreme)
        vtype = rng.integers(1, 25, n).astype(np.int32)
        nrad = np.ones(n, dtype=np.int32) if extreme else np.where(rng.random(n) < 0.1, 0, 1).astype(np.int32)
        out_a = np.full((n, 2), -7.0)
        out_b = out_a.copy()
        err = np.zeros(n, dtype=np.uint32)
        threw = np.zeros(n, dtype=np.int32)
        O.psn_counters(reset=True)
        L.lib.elmo_photosynthesis_batch(n, tp, vtype.ctypes.data, nrad.ctypes.data, x.ctypes.data, out_a.ctypes.data, err.ctypes.data)
        for k, v in O.psn_counters().items():
            total[k] += v
        L.ref_canopy.elmref_photosynthesis(n, tp, vtype.ctypes.data, nrad.ctypes.data, x.ctypes.data, out_b.ctypes.data, threw.ctypes.data)
        assert not threw.any() and not err.any()
        assert np.array_equal(out_a.view(np.uint64), out_b.view(np.uint64)), (seed, int((out_a != out_b).sum()))
        assert (out_a[nrad == 0, 0] == -7.0).all()  # no canopy layer: ci_z untouched
        assert np.isfinite(out_a).all()
    assert total["c4"] > 100_000 and total["brent"] > 10_000 and total["itmax"] >= 1, total


def test_photosynthesis_throw_sites_are_flagged_exactly_where_the_reference_throws(capfd):
    """The reference's throw sites in photosynthesis that no model state reaches, reached through
    photosynthesis() alone with inputs outside physics:
      * photosynthesis_impl.hh:232 "Negative stomatal conductance": the larger Ball-Berry root is negative only when the
        humidity or pressure term of the quadratic is - a NEGATIVE vapour pressure of the canopy air (eair < 0), a negative
        atmospheric pressure, a negative Ball-Berry slope in the plant-type row;
      * :289 "quadratic solution a == 0.0": a plant-type row with theta_cj = 0 (the co-limitation quadratic's leading coefficient).
    The reference throws in exactly the calls in which the restatement raises ELMO_ERR_PSN_NEG_GS / ELMO_ERR_PSN_QUADRATIC, and
    every call that does not throw returns the restatement's bits.  (:439, Brent's bracket check, is unreachable by construction:
    its only caller, hybrid(), calls brent() inside `if ((f1 < 0 && f0 > 0) || (f1 > 0 && f0 < 0))` (:589-596) with exactly those two
    values.  surface_albedo_impl.hh:270 cannot fire with nlevcan() == 1 (elm_constants.h:89): canopy_layer_lai has just set tlai_z(0)
    = elai and tsai_z(0) = esai (:226-229), so both differences are 0 or NaN, never > mpe; :306 sits inside `if (nlevcan() > 1)`.)"""
    NEG_GS, QUAD = 1 << 2, 1 << 3
    S = O.OracleState(1)
    S.load_params()
    L = O.lib()
    table = L.lib.elmo_pft_psn_ptr
    table.restype = C.c_void_p
    table.argtypes = [C.c_void_p]
    L.lib.elmo_photosynthesis_batch.argtypes = [C.c_int64] + [C.c_void_p] * 6
    tab = np.ctypeslib.as_array(C.cast(table(S.ptr), C.POINTER(C.c_double)), shape=(25, 27)).copy()
    n = 100_000
    rng = np.random.default_rng(4242)
    vtype = rng.integers(1, 25, n).astype(np.int32)
    nrad = np.ones(n, dtype=np.int32)

    def run(x, t):
        a = np.full((n, 2), -7.0)
        b = a.copy()
        err = np.zeros(n, dtype=np.uint32)
        threw = np.zeros(n, dtype=np.int32)
        t = np.ascontiguousarray(t)
        L.lib.elmo_photosynthesis_batch(n, t.ctypes.data, vtype.ctypes.data, nrad.ctypes.data, x.ctypes.data, a.ctypes.data, err.ctypes.data)
        L.ref_canopy.elmref_photosynthesis(n, t.ctypes.data, vtype.ctypes.data, nrad.ctypes.data, x.ctypes.data, b.ctypes.data, threw.ctypes.data)
        same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        assert same[threw == 0].all()
        return err, threw != 0

    mbb_neg = tab.copy()
    mbb_neg[:, 17] = -mbb_neg[:, 17]  # mbbopt (pft_data.h:22)
    cases = [("eair < 0", 7, tab), ("forc_pbot < 0", 3, tab), ("mbbopt < 0", None, mbb_neg)]
    reached = {}
    for what, col, t in cases:
        x = _psn_inputs(rng, n, False)
        if col is not None:
            x[:, col] = -x[:, col]
        err, threw = run(x, t)
        assert np.array_equal(threw, (err & NEG_GS) != 0) and not (err & QUAD).any(), what
        reached[what] = int(threw.sum())
    assert reached["eair < 0"] > 5000 and reached["mbbopt < 0"] > 5000 and reached["forc_pbot < 0"] >= 1, reached
    theta0 = tab.copy()
    theta0[:, 15] = 0.0  # theta_cj
    err, threw = run(_psn_inputs(rng, n, False), theta0)
    assert np.array_equal(threw, (err & QUAD) != 0) and threw.sum() > 50_000 and not (err & NEG_GS).any()
    C.CDLL(None).fflush(None)
    capfd.readouterr()  # (the reference prints its Ball-Berry check to stdout for thousands of these inputs, :240)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libelmref.so not built here")
def test_seven_wrappers_chained_by_the_reference():
    """The whole hot path - ELMInterface::advance's seven calls in its order (elm_kokkos_interface.cc:287-307) - run for four
    chained steps by NOTHING BUT the reference's own functions on one state and by the restatement on a copy: bit-identical
    after every wrapper of every step (no re-synchronisation between steps, so an error anywhere would compound)."""
    A = _state(8000, 31, every_pft=True)
    B = A.clone()
    R = O.Reference()
    hgt = {k: A[k].copy() for k in ("forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch")}
    for step in range(4):
        for k, v in hgt.items():  # the driver re-derives the forcing heights every step (atm_physics_impl.hh:197-203)
            A[k][...] = v
            B[k][...] = v
        for name, run_a, run_b in (
            ("frac_wet", A.frac_wet, lambda: R.frac_wet(B)),
            ("albedo_snicar", A.albedo_snicar, B.albedo_snicar_ref),
            ("canopy_hydrology", lambda: A.canopy_hydrology(DT), lambda: R.canopy_hydrology(B, DT)),
            ("surface_radiation", A.surface_radiation, lambda: R.surface_radiation(B)),
            ("canopy_temperature", A.canopy_temperature, lambda: R.canopy_temperature(B)),
            ("bareground_fluxes", A.bareground_fluxes, lambda: R.bareground_fluxes(B)),
            ("canopy_fluxes", lambda: A.canopy_fluxes(DT), lambda: B.canopy_fluxes_ref(DT)),
        ):
            run_a()
            run_b()
            assert not (B["err_flags"] & REF_THREW).any(), (step, name)
            d = _diff(A, B)
            assert not d, (step, name, d)
    assert (A["t_veg"] != A["forc_tbot"]).any() and (A["snl"] > 0).any()


def test_reference_functions_on_the_references_own_fixtures():
    """The reference's CanopyFluxes and SurfaceAlbedo fixtures (test/data/*_IN.txt, committed as tests/golden/*.npz), driven as
    test_CanFlux.cc / test_SurfAlb.cc drive them, through the reference's own functions and through the restatement: the same
    bits in every field - so the 73 comparisons in which the restatement misses CanopyFluxes_OUT.txt at 1e-15
    (tests/test_oracle_golden.py; BASELINE.md section 2 reports the same count for the compiled reference) are, value for
    value, the reference's own."""
    d = F.load("CanopyFluxes")
    rows = F.select_steps(d, "CanopyFluxes")
    probe = O.OracleState(1)
    fin, oin = F.split(d, "in/", rows, probe.nlev)
    fout, _ = F.split(d, "out/", rows, probe.nlev)
    beyond = 0
    for i in range(len(rows)):
        S = O.OracleState(1)
        S.load_params()
        S.set_scalars(**F.TEST_LAND, dayl=float(oin["dayl"][i, 0]), max_dayl=float(oin["max_dayl"][i, 0]))
        F.fill_state(S, {k: v[i : i + 1] for k, v in fin.items()})
        S["vtype"][:] = F.TEST_LAND["vtype"]
        B = S.clone()
        S.canopy_fluxes_given(F.TEST_DTIME, oin["forc_rho"][i], oin["forc_po2"][i], oin["forc_pco2"][i])
        B.canopy_fluxes_ref(F.TEST_DTIME, oin["forc_rho"][i], oin["forc_po2"][i], oin["forc_pco2"][i])
        assert not (B["err_flags"] & REF_THREW).any()
        dd = _diff(S, B)
        assert not dd, (i, dd)
        for name, exp in fout.items():
            ok = F.almost_equal(B[name].reshape(1, -1).astype(np.float64), exp[i : i + 1]) | np.isnan(exp[i : i + 1])
            beyond += int((~ok).sum())
    assert 0 < beyond <= 120, beyond  # the reference itself misses its fixture at 1e-15 in a few dozen values (73 when recorded)

    d = F.load("SurfaceAlbedo")
    rows = F.select_steps(d, "SurfaceAlbedo")
    S = O.OracleState(len(rows))
    S.load_params()
    S.set_scalars(**F.TEST_LAND)
    fin, _ = F.split(d, "in/", rows, S.nlev)
    F.fill_state(S, fin)
    S["vtype"][:] = F.TEST_LAND["vtype"]
    B = S.clone()
    sa = S.albedo_snicar_ex()
    sb = B.albedo_snicar_ref()
    assert not _diff(S, B) and np.array_equal(sa[0], sb[0], equal_nan=True) and np.array_equal(sa[1], sb[1], equal_nan=True)


def test_reference_throws_where_the_restatement_flags():
    """A throw site of the albedo wrapper that inputs can reach: a snow grain radius beyond the Mie table
    (snow_snicar_impl.hh:74-78).  The reference throws in exactly the columns in which the restatement raises
    ELMO_ERR_SNICAR_RDS, and every other column comes out bit for bit.  (The device raises the same bits as the restatement:
    tests/test_gpu_parity.py::test_error_flags_match_the_reference_throw_sites.)"""
    A = _state(4096, 91)
    bad = np.random.default_rng(5).choice(4096, 400, replace=False)
    A["snw_rds"][bad] = 5000.0
    A.frac_wet()
    B = A.clone()
    A.albedo_snicar()
    B.albedo_snicar_ref()
    threw = (B["err_flags"] & REF_THREW) != 0
    flagged = (A["err_flags"] & np.uint32(1 << 6)) != 0
    assert np.array_equal(threw, flagged) and 20 <= int(threw.sum()) < 400  # (only sunlit columns with a snow pack get there)
    keep = ~threw
    for k in A.fields:
        if k == "err_flags":
            continue
        a, b = A.fields[k][keep], B.fields[k][keep]
        eq = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
        assert eq.all(), (k, int((~eq).sum()))
