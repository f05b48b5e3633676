"""The oracle against the reference's SECOND fixture dump, test/new_data (committed as tests/golden/newdata/*.npz by
tests/golden/make_golden.py): the same ELM single-column site (US-Brw, vtype 12) under snow-free summer forcing, 97 steps per
module, 95 for SurfaceAlbedo.  No test of the reference reads these files, so they are golden vectors of ELM itself that
the restatement was never fitted to.  They add pinned branches the first dump does not have: SurfaceAlbedo with no snow at
all on every step and the sun up on every step (the first dump has a thin snow cover on every step and half the steps dark),
CanopyFluxes / CanopyTemperature / CanopyHydrology in warm, unfrozen conditions.

Same drivers and the same comparison (IsAlmostEqual rel 1e-15 / abs 1e-20) as tests/test_oracle_golden.py.
new_data/BareGroundFluxes_IN.txt is malformed (duplicate keys) and is not used.
"""
import numpy as np

from tests import fixtures as F
from tests import test_oracle_golden as T


def test_newdata_canopy_hydrology():
    with F.dataset("newdata"):
        d, rows, S, oin, fout = T._prepare("CanopyHydrology")
        assert len(rows) == 97
        S.set_scalars(oldfflag=int(oin["oldfflag"][0, 0]), dewmx=float(oin["dewmx"][0, 0]))
        S.canopy_hydrology(F.TEST_DTIME)
        S.frac_wet()
        total, bad = T._compare(S, fout, len(rows))
    assert total >= 3600 and not bad, bad
    assert not S["err_flags"].any()


def test_newdata_sunshade_and_surface_radiation():
    for module, minimum in (("CanopySunShadeFractions", 500), ("SurfaceRadiation", 2000)):
        with F.dataset("newdata"):
            d, rows, S, oin, fout = T._prepare(module)
            assert len(rows) == 97
            S.surface_radiation()
            total, bad = T._compare(S, fout, len(rows))
        assert total >= minimum and not bad, (module, bad)


def test_newdata_canopy_temperature():
    with F.dataset("newdata"):
        d, rows, S, oin, fout = T._prepare("CanopyTemperature")
        S.z0mr[:17] = oin["z0mr"][0]
        S.displar[:17] = oin["displar"][0]
        for k in "utq":
            S[f"forc_hgt_{k}_patch"][:] = oin[f"forc_hgt_{k}"][:, 0]
        S.canopy_temperature()
        total, bad = T._compare(S, fout, len(rows))
    assert total >= 22000 and not bad, bad
    assert (S["t_grnd"] > 273.15).all()  # the warm season: no frozen ground anywhere in this dump


def test_newdata_surface_albedo_snow_free():
    with F.dataset("newdata"):
        d, rows, S, oin, fout = T._prepare("SurfaceAlbedo")
        S.albsat[:] = oin["albsat"][0]
        S.albdry[:] = oin["albdry"][0]
        S["isoicol"][:] = 3
        sun, sha = S.albedo_snicar_ex()
        total, bad = T._compare(S, fout, len(rows), skip=("fabd_sun", "fabd_sha"))
        for name, got in (("fabd_sun", sun), ("fabd_sha", sha)):
            assert (F.almost_equal(got, fout[name]) | np.isnan(fout[name])).all(), name
    assert total >= 15000 and not bad, bad
    assert not S["err_flags"].any()
    # the branches this dump pins and the first one does not: no snow at all, the sun up on every step
    assert (S["h2osno"] == 0).all() and (S["coszen"] > 0).all()


def test_newdata_canopy_fluxes():
    with F.dataset("newdata"):
        steps, got, fout, niters, flags = T._run_canflux(given=True)
    total, nloose, worst = 0, 0, 0.0
    for name, exp in fout.items():
        ok = F.almost_equal(got[name], exp) | np.isnan(exp)
        total += ok.size
        nloose += int((~ok).sum())
        if not ok.all():
            # (relative to the field's own scale: the values that are not 1e-15-equal are differences of 1e-19 .. 1e-12)
            scale = max(float(np.nanmax(np.abs(exp))), 1e-300)
            worst = max(worst, float(np.max(np.where(ok, 0.0, np.abs(got[name] - exp))) / scale))
    # like the first dump (where the compiled reference itself leaves 73 values beyond 1e-15, worst 3.5e-10): a bounded set
    assert total >= 23000
    assert nloose <= 150, nloose
    assert worst < 1e-9, worst
    assert (flags & 0x7FF) == 0
    assert niters.min() >= 3 and niters.max() <= 41
