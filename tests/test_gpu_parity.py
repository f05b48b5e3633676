"""HIP kernels (through the C ABI) vs the oracle on identical inputs.  Run on the GPU box: pytest -m gpu.

Bar (BASELINE.json north_star / SURVEY.md 8(c)): fp64 outputs within 1e-12 relative (+1e-18 absolute floor for
catastrophic-cancellation outputs) of the CPU path; integer fields and error flags exact.
"""
import numpy as np
import pytest

from elmkernels_amd import state as st
from elmkernels_amd import synth
from tests import fixtures as F
from tests import helpers as H

pytestmark = pytest.mark.gpu

DT = 1800.0


def _pair(n, tier, seed, land=None):
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=seed)
    S = H.oracle_state(cols, scal, soil, land)
    D = H.device_state(cols, scal, soil, land)
    return D, S


def _check(D, S, what, rel=H.REL_TOL, names=None, skip_cols=None, bitwise=False):
    worst, bad = H.compare_states(D, S, names=names, rel=rel, skip_cols=skip_cols, bitwise=bitwise)
    assert not bad, f"{what}: worst rel err {worst:.3e}; over tolerance: {bad}"
    return worst


def test_roundtrip_layouts():
    """upload/download in both host layouts reproduce the data (level transposition through LDS tiles)."""
    D = st.ELMState(1000)
    rng = np.random.default_rng(0)
    for name in ("t_soisno", "zisoi", "albd", "snl", "veg_active", "forc_tbot", "snw_rds"):
        fid, nlev, dt = D.fields[name]
        a = (rng.random((1000, nlev)) * 100).astype(dt)
        if name == "snl":
            a %= 6  # (elmk_upload refuses a number of snow layers outside 0..5)
        D.upload(name, a)
        assert np.array_equal(D.download(name).reshape(1000, nlev), a), name
        soa = D.download(name, layout=st.LAYOUT_SOA).reshape(nlev, 1000)
        assert np.array_equal(soa, a.T), name
        D.upload(name, np.ascontiguousarray(a.T[:, 100:300]), col0=100, layout=st.LAYOUT_SOA)
        assert np.array_equal(D.download(name, col0=100, n=200).reshape(200, nlev), a[100:300]), name
    D.close()


@pytest.mark.parametrize("tier,n,seed", [("A", 47, 1), ("A", 5000, 2), ("B", 3008, 3), ("B", 20000, 4)])
def test_each_wrapper_in_timestep_order(tier, n, seed):
    """Each of the seven wrappers, checked right after it runs, so errors cannot hide behind later kernels.
    Before each wrapper the device state is re-synchronised from the oracle: every kernel is tested on
    bit-identical inputs, and its outputs must be bit-identical (helpers.LIBM_RESIDUAL_FIELDS: within 1e-12)."""
    D, S = _pair(n, tier, seed)
    calls = [
        ("frac_wet", lambda: st.kokkos_frac_wet(D), S.frac_wet),
        ("albedo_snicar", lambda: st.kokkos_albedo_snicar(D), S.albedo_snicar),
        ("canopy_hydrology", lambda: st.kokkos_canopy_hydrology(D, DT), lambda: S.canopy_hydrology(DT)),
        ("surface_radiation", lambda: st.kokkos_surface_radiation(D), S.surface_radiation),
        ("canopy_temperature", lambda: st.kokkos_canopy_temperature(D), S.canopy_temperature),
        ("bareground_fluxes", lambda: st.kokkos_bareground_fluxes(D), S.bareground_fluxes),
        ("canopy_fluxes", lambda: st.kokkos_canopy_fluxes(D, DT), lambda: S.canopy_fluxes(DT)),
    ]
    for name, dev, ora in calls:
        dev()
        ora()
        flags_d = D["err_flags"]
        flags_o = S["err_flags"]
        fatal = ((flags_d | flags_o) & 0x7FF) != 0
        assert np.array_equal(flags_d & 0x7FF, flags_o & 0x7FF), f"{name}: fatal flag sets differ"
        _check(D, S, f"{tier}/{n}/{name}", skip_cols=fatal if fatal.any() else None, bitwise=True)
        for k, v in S.fields.items():  # re-sync: next kernel starts from identical bits
            if k != "err_flags":
                D[k] = v
    D.close()


@pytest.mark.parametrize("tier,n,seed", [("A", 4700, 11), ("B", 12000, 12)])
def test_full_timestep_chain(tier, n, seed):
    """elmk_timestep7 (no re-sync between kernels) for three consecutive steps vs the oracle chain: the wiring of the
    whole step (kernel order, state flow between kernels and between steps).  Every kernel is bit-exact on bit-identical
    inputs, so the chain stays bit-identical: every field, every column, every step."""
    D, S = _pair(n, tier, seed)
    for step in range(3):
        st.timestep7(D, DT)
        S.timestep7(DT)
        _check(D, S, f"{tier}/{n}/step {step}", bitwise=True)
        assert np.array_equal(D["err_flags"] & 0x7FF, S["err_flags"] & 0x7FF)
        flags, first = D.error_summary()
        assert (flags & 0x7FF) == int(np.bitwise_or.reduce(S["err_flags"]) & 0x7FF)
    D.close()


def test_advance_chain_bit_identical():
    """The reference's advance() order past the seven wrappers - init_timestep, the seven, soil_temperature,
    surface_fluxes - chained for two steps with no re-synchronisation: still bit-identical in every field."""
    n = 6016
    D, S = _pair(n, "B", 13)
    for step in range(2):
        st.kokkos_init_timestep(D)
        S.init_timestep()
        st.timestep7(D, DT)
        S.timestep7(DT)
        st.kokkos_soil_temperature(D, DT)
        S.soil_temperature(DT)
        st.kokkos_surface_fluxes(D, DT)
        S.surface_fluxes(DT)
        _check(D, S, f"advance chain step {step}", bitwise=True)
    D.close()


@pytest.mark.parametrize("tier,n,seed", [("A", 4700, 41), ("B", 20000, 42), ("B", 1001, 43)])
def test_snow_hydrology_next_row(tier, n, seed):
    """kokkos_snow_hydrology (snow_hydrology_kokkos.cc:23-188) on bit-identical inputs: the state after the seven wrappers and
    the temperature solve (so that imelt, swe_old, frac_iceold, qflx_snomelt ... are what the step produced).  Every field
    bit-identical to the oracle's restatement, the new flag bits included (what pins the oracle for this row: oracle/elmo_physics_g.c)."""
    D, S = _pair(n, tier, seed)
    S.timestep7(DT)
    S.soil_temperature(DT)
    for k, v in S.fields.items():
        if k != "err_flags":
            D[k] = v
    snl0 = S["snl"].copy()
    st.kokkos_snow_hydrology(D, DT)
    S.snow_hydrology(DT)
    _check(D, S, f"snow_hydrology {tier}/{n}", bitwise=True)
    assert np.array_equal(D["err_flags"], S["err_flags"] & np.uint32(0xF000))  # (the device flags were clear before the call)
    if tier == "B":
        assert (S["snl"] != snl0).sum() > n // 100 and (S["err_flags"] & (1 << 12)).any()
    D.close()


def test_snow_hydrology_other_land_units():
    """combine_layers treats soil / crop / urban and wetland / land-ice units differently (where the water of a vanishing
    layer goes, snow_hydrology_impl.hh:675-720, :760-775)."""
    for land in (dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0), dict(ltype=3, ctype=0, vtype=0, urbpoi=0, lakpoi=0),
                 dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0), dict(ltype=2, ctype=0, vtype=15, urbpoi=0, lakpoi=0)):
        D, S = _pair(3000, "B", 44, land)
        S["h2osoi_ice"][::7, :5] *= 0.001  # thin-ice layers: the first loop of combine_layers
        for k, v in S.fields.items():
            if k != "err_flags":
                D[k] = v
        st.kokkos_snow_hydrology(D, DT)
        S.snow_hydrology(DT)
        _check(D, S, f"snow_hydrology, land {land}", bitwise=True)
        D.close()


def test_advance_chain_with_snow_hydrology():
    """The reference's whole advance() order (elm_kokkos_interface.cc:278-318) - init_timestep, the seven wrappers,
    soil_temperature, snow_hydrology, surface_fluxes (per wrapper, with the fused seven, or as the single call
    elmk_advance_physics, plain and as a replayed HIP graph) - chained for twelve steps with no re-synchronisation: the snow pack
    is re-meshed on the device step after step (layers appear, merge, split, vanish) and every field stays bit-identical
    to the oracle chain.  The forcing heights are put back before each step, as the driver does (atm_physics_impl.hh:197-203)."""
    n = 6016
    D, S = _pair(n, "B", 14)
    hgt = {k: S[k].copy() for k in ("forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch")}
    snl_hist = []
    for step in range(12):
        for k, v in hgt.items():
            D[k] = v
            S[k][...] = v
        st.kokkos_init_timestep(D)
        S.init_timestep()
        if step % 3 == 2:  # the whole device part of advance() as one call; as one replayed HIP graph in the second half
            D.set_graph(step >= 6)
            st.advance_physics(D, DT)
            D.set_graph(False)
        else:
            (st.timestep7_fused if step % 2 else st.timestep7)(D, DT)
            st.kokkos_soil_temperature(D, DT)
            st.kokkos_snow_hydrology(D, DT)
            st.kokkos_surface_fluxes(D, DT)
        S.timestep7(DT)
        S.soil_temperature(DT)
        S.snow_hydrology(DT)
        S.surface_fluxes(DT)
        _check(D, S, f"advance chain with snow hydrology, step {step}", bitwise=True)
        snl_hist.append(S["snl"].copy())
    changed = sum(int((a != b).sum()) for a, b in zip(snl_hist, snl_hist[1:]))
    assert changed > 50  # the mesh really moved between steps
    D.close()



def test_cpp_interface_mirror(tmp_path):
    """include/elmk_interface.hpp - the C++ mirror of the reference's driver class ELM::ELMInterface (setup, advance,
    getPrimaryVars; elm_kokkos_interface.cc:38-358) - compiled with g++ against libelmk and run as the reference's driver
    loop (examples/elm_interface_demo.cc): three advance() steps (phenology, forcing, init_timestep, the ten physics calls
    as one HIP graph, conservation) give the oracle chain's PrimaryVars and conservation diagnostics bit for bit."""
    import shutil
    import struct
    import subprocess

    from elmkernels_amd import _lib as L

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    n, nsteps = 3008, 3
    D, S = _pair(n, "B", 33)
    D.close()
    exe = tmp_path / "demo"
    import os

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "elm_interface_demo.cc"), "-L" + libdir, "-lelmk",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    rng = np.random.default_rng(3)
    e = rng.random(8)
    wt1, wt2 = 1.0 - e, e
    blob = [struct.pack("<q", n)]

    def rec(name, kind, arr):
        a = np.ascontiguousarray(arr)
        blob.append(name.encode().ljust(32, b"\0") + struct.pack("<iq", kind, a.nbytes) + a.tobytes())

    for k, v in S.fields.items():
        if k != "err_flags":
            rec(k, 0, v)
    sc = S.scalars
    rec("land", 1, np.array([sc["ltype"], sc["ctype"], sc["vtype"], sc["urbpoi"], sc["lakpoi"]], np.int32))
    rec("scalars", 1, np.array([sc["dewmx"], sc["oldfflag"], sc["dayl"], sc["max_dayl"], DT], np.float64))
    rec("pft_psn", 1, S.pft_psn)
    rec("pft_alb", 1, S.pft_alb)
    rec("z0mr", 1, S.z0mr)
    rec("displar", 1, S.displar)
    rec("albsat", 1, S.albsat)
    rec("albdry", 1, S.albdry)
    for i, name in enumerate(L.SNICAR_NAMES):
        rec(f"snicar/{i}", 1, S.snicar[name])
    rec("age_tau", 1, S.snowage[0])
    rec("age_kappa", 1, S.snowage[1])
    rec("age_drdt0", 1, S.snowage[2])
    rec("forc_wt1", 1, wt1)
    rec("forc_wt2", 1, wt2)
    rec("month_wt", 1, np.array([0.3, 0.7]))
    (tmp_path / "state.bin").write_bytes(b"".join(blob))
    out = subprocess.run([str(exe), str(tmp_path / "state.bin"), str(nsteps), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for _ in range(nsteps):
        S.phenology(0.3, 0.7)
        S.get_forcing(wt1, wt2, False)
        S.init_timestep()
        S.timestep7(DT)
        S.soil_temperature(DT)
        S.snow_hydrology(DT)
        S.surface_fluxes(DT)
        diag = S.evaluate_conservation(DT)
    raw = (tmp_path / "out.bin").read_bytes()
    off = 0

    def same_bits(a, b):
        return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))

    for name, dt_, count in (("t_soisno", np.float64, n * 20), ("h2osoi_liq", np.float64, n * 20), ("h2osoi_ice", np.float64, n * 20),
                             ("t_grnd", np.float64, n), ("h2osno", np.float64, n), ("snl", np.int32, n)):
        got = np.frombuffer(raw, dt_, count, off).reshape(S[name].shape)
        off += got.nbytes
        assert same_bits(got, S[name]), name
    got = np.frombuffer(raw, np.float64, 24, off).reshape(8, 3)
    assert same_bits(got[:, 0], diag.min(axis=0)) and same_bits(got[:, 1], diag.max(axis=0))
    np.testing.assert_allclose(got[:, 2], diag.sum(axis=0), rtol=1e-9, atol=1e-6)  # (sums: association order)



def test_initialize_state_then_advance():
    """ELM::initialize_kokkos_elm's per-column init functions (initialize_elm_kokkos.cc:373-428) on the device - every land
    unit, every branch of the initial snow mesh, mineral to pure organic soil - bit-identical to the oracle, which is pinned
    bit for bit against the reference's own headers (tests/test_oracle_vs_ref.py::test_initialize_state_bitwise); then the
    cold-start state goes straight into two advance() steps on the device, still bit-identical."""
    pft, _ = synth.load_params()
    lands = [dict(ltype=1, ctype=0, vtype=2, urbpoi=0, lakpoi=0), dict(ltype=4, ctype=0, vtype=0, urbpoi=0, lakpoi=0),
             dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0), dict(ltype=5, ctype=0, vtype=0, urbpoi=0, lakpoi=1),
             dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0), dict(ltype=7, ctype=75, vtype=0, urbpoi=1, lakpoi=0)]
    for k, land in enumerate(lands):
        n = 6016
        ft = st.field_table()
        cols, scal, soil = synth.make_state(ft, n, tier="B", seed=400 + k)
        cols["snow_depth"] = synth.init_snow_depths(n, 400 + k)
        vt = cols["vtype"].copy()
        vt[::9] = 0
        cols["vtype"] = vt
        S = H.oracle_state(cols, scal, soil, land)
        D = H.device_state(cols, scal, soil, land)
        S.set_init_params(synth.ORGANIC_MAX, pft["roota_par"], pft["rootb_par"])
        D.set_init_params(synth.ORGANIC_MAX, pft["roota_par"], pft["rootb_par"])
        st.initialize_kokkos_elm(D)
        S.initialize_state()
        _check(D, S, f"initialize_state, land {land}", bitwise=True)
        assert set(np.unique(S["snl"]).tolist()) == ({0} if land["lakpoi"] else {0, 1, 2, 3, 4, 5})
        if k == 0:  # soil columns: the freshly initialised state through the whole advance() order, twice
            for step in range(2):
                st.kokkos_init_timestep(D)
                S.init_timestep()
                st.advance_physics(D, DT)
                S.timestep7(DT)
                S.soil_temperature(DT)
                S.snow_hydrology(DT)
                S.surface_fluxes(DT)
                _check(D, S, f"advance() step {step} from the cold-start state", bitwise=True)
        D.close()



def test_python_interface_mirror():
    """elmkernels_amd.ELMInterface - the Python counterpart of include/elmk_interface.hpp, mirroring ELM::ELMInterface's setup /
    advance / getPrimaryVars - from a cold start: initialize(), then two advance() steps (phenology, forcing, init_timestep,
    the ten physics calls as one HIP graph, conservation) equal the oracle chain bit for bit in every PrimaryVars member."""
    import elmkernels_amd as E

    n = 3008
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=55)
    cols["snow_depth"] = synth.init_snow_depths(n, 55)
    pft, optics = synth.load_params()
    S = H.oracle_state(cols, scal, soil)
    S.set_init_params(synth.ORGANIC_MAX, pft["roota_par"], pft["rootb_par"])
    elm = E.ELMInterface(n)
    elm.setup(synth.TEST_LAND, scal, pft, optics, (soil["albsat"], soil["albdry"]), synth.snow_age_tables(),
              init_params=(synth.ORGANIC_MAX, pft["roota_par"], pft["rootb_par"]))
    for k, v in cols.items():
        elm.S[k] = v
    elm.initialize()
    S.initialize_state()
    e = np.random.default_rng(9).random(8)
    for _ in range(2):
        assert elm.advance(DT, 1.0 - e, e, 0.3, 0.7) is False
        S.phenology(0.3, 0.7)
        S.get_forcing(1.0 - e, e, False)
        S.init_timestep()
        S.timestep7(DT)
        S.soil_temperature(DT)
        S.snow_hydrology(DT)
        S.surface_fluxes(DT)
    _check(elm.S, S, "python ELMInterface: two advance() steps from a cold start", bitwise=True)
    pv = elm.getPrimaryVars()
    assert set(pv) == set(E.ELMInterface.PRIMARY_VARS)
    for k, got in pv.items():
        ref = S[k].astype(got.dtype)
        if got.dtype.kind == "f":  # (any NaN equals any NaN: a cold start with layers but no snow mass divides 0 by 0, as the reference does)
            assert ((got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))).all(), k
        else:
            assert np.array_equal(got, ref), k
    assert elm.conservation.shape == (8, 3)
    elm.close()


def test_other_land_units():
    """Non-soil land units take the short branches of every routine (wetland, land ice, lake, urban)."""
    for land in (dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0), dict(ltype=3, ctype=0, vtype=0, urbpoi=0, lakpoi=0),
                 dict(ltype=5, ctype=0, vtype=0, urbpoi=0, lakpoi=1), dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0)):
        D, S = _pair(2000, "B", 21, land)
        st.timestep7(D, DT)
        S.timestep7(DT)
        _check(D, S, f"land {land}", bitwise=True)
        D.close()


def test_fixture_steps_on_device():
    """The reference's own fixture inputs (one step per column), the five streaming modules through the HIP kernels:
    HIP vs the ELM _OUT records with the reference's own comparison, ELM::IO::IsAlmostEqual (relative 1e-15 / absolute 1e-20,
    src/utils/read_test_input.hh:17-24) - the bar the reference's tests hold and the oracle meets on the CPU - and HIP vs the
    oracle on the same steps bit for bit."""
    from oracle import oracle as O

    pft, optics = synth.load_params()
    for module, run, run_oracle in (
        ("CanopyHydrology", lambda D: (st.kokkos_canopy_hydrology(D, DT), st.kokkos_frac_wet(D)),
         lambda S: (S.canopy_hydrology(DT), S.frac_wet())),
        ("SurfaceRadiation", lambda D: st.kokkos_surface_radiation(D), lambda S: S.surface_radiation()),
        ("CanopySunShadeFractions", lambda D: st.kokkos_surface_radiation(D), lambda S: S.surface_radiation()),
        ("CanopyTemperature", lambda D: st.kokkos_canopy_temperature(D), lambda S: S.canopy_temperature()),
        ("SurfaceAlbedo", lambda D: st.kokkos_albedo_snicar(D), lambda S: S.albedo_snicar()),
    ):
        d = F.load(module)
        rows = F.select_steps(d, module)
        D = st.ELMState(len(rows))
        D.set_pft(pft)
        D.set_snicar(optics)
        D.set_land(**F.TEST_LAND)
        S = O.OracleState(len(rows))
        S.load_params(pft, optics)
        S.set_scalars(**F.TEST_LAND)
        nlev = {k: v[1] for k, v in D.fields.items()}
        fin, oin = F.split(d, "in/", rows, nlev)
        fout, _ = F.split(d, "out/", rows, nlev)
        F.fill_state(S, fin)
        for k, v in fin.items():
            D[k] = np.nan_to_num(v, nan=0.0) if D.fields[k][2] != np.float64 else v
        D["vtype"] = np.full(len(rows), 12, np.int32)
        D["veg_active"] = np.ones(len(rows), np.uint8)
        S["vtype"][:] = 12
        S["veg_active"][:] = 1
        if module == "CanopyHydrology":
            sc = dict(oldfflag=int(oin["oldfflag"][0, 0]), dewmx=float(oin["dewmx"][0, 0]))
            D.set_scalars(**sc)
            S.set_scalars(**sc)
        if module == "CanopyTemperature":
            for k in "utq":
                D[f"forc_hgt_{k}_patch"] = oin[f"forc_hgt_{k}"][:, 0]
                S[f"forc_hgt_{k}_patch"][:] = oin[f"forc_hgt_{k}"][:, 0]
        if module == "SurfaceAlbedo":
            D.set_soilcolor(np.tile(oin["albsat"][0], (20, 1)), np.tile(oin["albdry"][0], (20, 1)))
            D["isoicol"] = np.full(len(rows), 3, np.int32)
            S.albsat[:] = oin["albsat"][0]
            S.albdry[:] = oin["albdry"][0]
            S["isoicol"][:] = 3
        run(D)
        run_oracle(S)
        total = 0
        for name, exp in fout.items():
            if module == "SurfaceAlbedo" and name in ("fabd_sun", "fabd_sha"):
                continue  # wrapper-local in the reference, never stored in the state
            got = D[name].reshape(len(rows), -1).astype(np.float64)
            ok = F.almost_equal(got, exp) | np.isnan(exp)
            total += ok.size
            assert ok.all(), f"{module}.{name}: {int((~ok).sum())} values beyond IsAlmostEqual, worst {float(np.where(ok, 0.0, F.rel_err(got, exp)).max()):.3e}"
        assert total >= 700, (module, total)
        # ... and the device against the oracle on the same steps: every field the module writes, bit for bit
        worst, bad = H.compare_states(D, S, names=[n for n in fout if n in S.fields and n in D.fields], bitwise=True)
        assert not bad, f"{module}: HIP vs oracle {bad}"
        D.close()


def test_the_gpu_run_holds_the_bitwise_bar():
    """On the GPU selection the device-vs-oracle comparisons must not silently fall back from bit identity to the 1e-12
    tolerance (a missing gcc, a failed probe build or another host libm would do that, and the suite would still be green).
    ELMK_ALLOW_TOLERANCE=1 opts in to the fallback explicitly (a GPU host with another libm)."""
    import os

    from tests import _parity_mode as M

    print("elmk parity bar on this host:", M.REASON)
    if os.environ.get("ELMK_ALLOW_TOLERANCE") != "1":
        assert M.BITWISE_VALID, M.REASON
    # ... nor may the two HIP-vs-reference-library tests silently skip: oracle/_ref/*.so are git-ignored binaries that cannot be
    # rebuilt on the GPU box (/root/reference does not exist there) - a run from a clean checkout would go green without them.
    # `python -c "import __graft_entry__ as g; g.build()"` in the build container makes them; ELMK_ALLOW_NO_REF=1 opts out.
    if os.environ.get("ELMK_ALLOW_NO_REF") != "1":
        from oracle import oracle as O

        L = O.lib()
        missing = [n for n, h in (("libelmref.so", L.ref), ("libelmref_canopy.so", L.ref_canopy), ("libelmref_soil.so", L.ref_soil),
                                  ("libelmref_snow.so", L.ref_snow)) if h is None]
        assert not missing, f"oracle/_ref/ lacks {missing}: the HIP-vs-reference-library tests would skip (ELMK_ALLOW_NO_REF=1 to allow)"


def test_snl_outside_its_range_is_refused_at_upload():
    """snl indexes the level arrays in every snow / soil kernel: a value outside 0..5 never reaches the device."""
    D = st.ELMState(64)
    bad = np.zeros(64, np.int32)
    bad[7] = 6
    with pytest.raises(RuntimeError):
        D["snl"] = bad
    bad[7] = -1
    with pytest.raises(RuntimeError):
        D["snl"] = bad
    D["snl"] = np.full(64, 5, np.int32)
    assert (D["snl"] == 5).all()
    D.close()


def test_fp32_state_variant_error_envelope():
    """BASELINE config 5's second half (libelmk_f32.so: every fp64 state field stored as fp32, fp64 arithmetic, fused step) is
    a REPORT, not a parity claim: its outputs cannot be within 1e-12 of the reference's.  What is asserted is the documented
    envelope of one step against the oracle on the same inputs: inputs are held to fp32 rounding, the
    median relative difference of the outputs stays below 1e-6, nine values in ten below 1e-5, and the int fields that do not
    depend on a tolerance-terminated iteration are exact.  (The tails are NOT small - a leaf-temperature iteration that takes
    one trip more or fewer moves a flux by per cent - which is why this build is never the product.)"""
    from elmkernels_amd import _lib as L

    n = 20000
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier="A", seed=77)
    D = st.ELMState(n, lib_path=L.F32_LIB_PATH)
    assert D.lib.elmk_state_real_bytes() == 4
    pft, optics = synth.load_params()
    D.set_pft(pft); D.set_snicar(optics); D.set_soilcolor(soil["albsat"], soil["albdry"])
    D.set_land(**synth.TEST_LAND); D.set_scalars(**scal)
    for k, v in cols.items():
        D.upload(k, v, col0=0)
    # storage really is fp32: what comes back is the input rounded to nearest fp32, and the state takes half the bytes
    back = D["t_soisno"]
    assert np.array_equal(back, cols["t_soisno"].astype(np.float32).astype(np.float64))
    D64 = st.ELMState(n)
    assert D.device_bytes < 0.75 * D64.device_bytes
    D64.close()
    S = H.oracle_state(cols, scal, soil)
    st.timestep7_fused(D, DT)
    S.timestep7(DT)
    rels = []
    for name in ("t_veg", "t_grnd", "h2ocan", "btran", "qflx_evap_tot", "eflx_sh_tot", "eflx_lh_tot", "cgrnd", "t_ref2m", "q_ref2m", "albd", "albi",
                 "fabd", "sabg", "sabv", "fsa", "fsr", "qg", "thm", "tssbef", "rootr", "eff_porosity", "fwet", "fdry", "ulrad", "dlrad"):
        a, b = D[name].astype(np.float64).ravel(), S[name].astype(np.float64).ravel()
        scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)
        r = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, np.abs(a - b) / scale)
        rels.append(r[np.isfinite(r)])
    r = np.concatenate(rels)
    med, p90 = float(np.median(r)), float(np.percentile(r, 90))
    print(f"fp32 state vs oracle after one step: median {med:.2e}  p90 {p90:.2e}  p99 {np.percentile(r, 99):.2e}  max {r.max():.2e}")
    assert med < 1e-6 and p90 < 1e-5, (med, p90)
    assert np.array_equal(D["snl"], S["snl"]) and np.array_equal(D["nrad"], S["nrad"])
    flags, _ = D.error_summary()
    assert (flags & 0x7FF) == 0
    D.close()


def test_hot_path_on_device_against_the_reference_library():
    """HIP against the REFERENCE ITSELF with no restatement in between: the prebuilt oracle/_ref/libelmref.so and
    libelmref_canopy.so (the reference's own physics headers, compiled in the build container; the binaries travel to the GPU
    box, /root/reference does not) run the seven wrappers of the step on the host - the oracle's state object is only the
    container of the arrays - and the device runs elmk_timestep7 / elmk_timestep7_fused on the same start state: three chained
    model steps, 40 000 branch-mix columns with all 24 leafed plant types (C3 and C4), then a soybean land unit and the crop,
    land-ice, deep-lake, wetland and urban land units: every field bit for bit."""
    from oracle import oracle as O
    from tests import _parity_mode

    if not (O.have_ref() and O.have_ref_canopy()):
        pytest.skip("oracle/_ref libraries not built")
    if not _parity_mode.BITWISE_VALID:
        pytest.skip("another host libm than the one the device math restates")
    R = O.Reference()
    for n, seed, land in ((40000, 71, None), (8000, 72, dict(ltype=1, ctype=1, vtype=23, urbpoi=0, lakpoi=0)),
                          (4000, 73, dict(ltype=2, ctype=0, vtype=15, urbpoi=0, lakpoi=0)),    # crop
                          (4000, 74, dict(ltype=3, ctype=0, vtype=0, urbpoi=0, lakpoi=0)),     # land ice
                          (4000, 75, dict(ltype=5, ctype=0, vtype=0, urbpoi=0, lakpoi=1)),     # deep lake
                          (4000, 76, dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0)),     # wetland
                          (4000, 77, dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0))):   # urban
        ft = st.field_table()
        cols, scal, soil = synth.make_state(ft, n, tier="B", seed=seed)
        cols["vtype"] = np.random.default_rng(seed).integers(1, 25, n).astype(np.int32)
        B = H.oracle_state(cols, scal, soil, land)
        D = H.device_state(cols, scal, soil, land)
        for step in range(3):
            (st.timestep7_fused if step == 1 else st.timestep7)(D, DT)
            R.frac_wet(B)
            B.albedo_snicar_ref()
            R.canopy_hydrology(B, DT)
            R.surface_radiation(B)
            R.canopy_temperature(B)
            R.bareground_fluxes(B)
            B.canopy_fluxes_ref(DT)
            assert not (B["err_flags"] >> 31).any(), "the reference threw"
            _check(D, B, f"HIP vs the reference's own functions, land {land}, step {step}", bitwise=True)
            assert (D["err_flags"] & 0x7FF == 0).all()
        D.close()


def test_advance_rows_on_device_against_the_reference_library():
    """The same for the rows that follow the seven in ELMInterface::advance and that the reference's own functions can run whole:
    kokkos_init_timestep's column functor, the seven wrappers, kokkos_soil_temperature and kokkos_surface_fluxes, by nothing
    but oracle/_ref on the host and by the HIP kernels on the device, two chained steps on 20 000 branch-mix columns: every
    field bit for bit.  (kokkos_snow_hydrology is left out of this chain: two of its ten stages have no reference run.)"""
    from oracle import oracle as O
    from tests import _parity_mode

    if not (O.have_ref() and O.have_ref_canopy() and O.lib().ref_soil is not None):
        pytest.skip("oracle/_ref libraries not built")
    if not _parity_mode.BITWISE_VALID:
        pytest.skip("another host libm than the one the device math restates")
    R = O.Reference()
    n = 20000
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=81)
    B = H.oracle_state(cols, scal, soil)
    D = H.device_state(cols, scal, soil)
    hgt = {k: B[k].copy() for k in ("forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch")}
    for step in range(2):
        for k, v in hgt.items():
            D[k] = v
            B[k][...] = v
        st.kokkos_init_timestep(D)
        st.timestep7(D, DT)
        st.kokkos_soil_temperature(D, DT)
        st.kokkos_surface_fluxes(D, DT)
        B.init_timestep(lib=R.R)
        R.frac_wet(B)
        B.albedo_snicar_ref()
        R.canopy_hydrology(B, DT)
        R.surface_radiation(B)
        R.canopy_temperature(B)
        R.bareground_fluxes(B)
        B.canopy_fluxes_ref(DT)
        B.soil_temperature_ref(DT)
        B.surface_fluxes(DT, lib=R.R)
        assert not (B["err_flags"] >> 31).any(), "the reference threw"
        _check(D, B, f"HIP vs the reference's own functions, advance rows, step {step}", bitwise=True)
    D.close()


def test_reference_outputs_on_device():
    """HIP directly against outputs of the REFERENCE ITSELF on branch-mix columns (tests/golden/ref_branch_mix.npz, recorded in
    the build container from oracle/_ref by tests/refgolden.py): init_timestep, frac_wet, canopy_hydrology,
    surface_radiation, canopy_temperature, bareground_fluxes, SNICAR's products, the whole soil_temperature solve,
    surface_fluxes and the conservation diagnostics.  The inputs of every stage are the oracle chain's state (re-synchronised
    before the stage; its hash must be the recorded one), the expected outputs are the reference's: bit for bit."""
    from tests import _parity_mode
    from tests import refgolden as G

    if not _parity_mode.BITWISE_VALID:
        pytest.skip("another host libm: the oracle chain is not the one the fixture was recorded on")
    fx = G.load()
    (cols, scal, soil), S = G.start_state()
    D = H.device_state(cols, scal, soil)
    dev = {"init_timestep": lambda: st.kokkos_init_timestep(D), "frac_wet": lambda: st.kokkos_frac_wet(D),
           "albedo_snicar": lambda: st.kokkos_albedo_snicar(D), "canopy_hydrology": lambda: st.kokkos_canopy_hydrology(D, G.DT),
           "surface_radiation": lambda: st.kokkos_surface_radiation(D), "canopy_temperature": lambda: st.kokkos_canopy_temperature(D),
           "bareground_fluxes": lambda: st.kokkos_bareground_fluxes(D), "canopy_fluxes": lambda: st.kokkos_canopy_fluxes(D, G.DT),
           "soil_temperature": lambda: st.kokkos_soil_temperature(D, G.DT), "snow_hydrology": lambda: st.kokkos_snow_hydrology(D, G.DT),
           "surface_fluxes": lambda: st.kokkos_surface_fluxes(D, G.DT)}
    checked = 0
    for stage in G.STAGES:
        assert str(fx[f"hash/{stage}"]) == G.state_hash(S), f"{stage}: the oracle chain is not the recorded one"
        for k, v in S.fields.items():  # the stage starts from the recorded inputs
            if k != "err_flags":
                D[k] = v
        day = S["coszen"] > 0
        dev[stage]()
        G.run_oracle(S, stage)
        for k in [k for k in fx.files if k.startswith(f"out/{stage}/")]:
            got, exp = D[k.split("/")[2]], fx[k]
            if exp.dtype.kind == "f":
                same = (got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))
            else:
                same = got == exp
            assert same.all(), f"{k}: {int((~same).sum())} values differ from the reference's"
            checked += 1
        if stage == "albedo_snicar":
            assert np.array_equal(D["albsnd"][day], fx["out/snicar/albsnd"][day]) and np.array_equal(D["albsni"][day], fx["out/snicar/albsni"][day])
            for name, f, band, alb in (("flx_absdv", "flx_absd_snw", 0, "albsnd"), ("flx_absdn", "flx_absd_snw", 1, "albsnd"),
                                       ("flx_absiv", "flx_absi_snw", 0, "albsni"), ("flx_absin", "flx_absi_snw", 1, "albsni")):
                exp = fx[f"out/snicar/{f}"][:, :, band] * (1.0 - fx[f"out/snicar/{alb}"][:, band : band + 1])
                assert np.array_equal(D[name][day], exp[day]), name
            checked += 6
    for k, v in S.fields.items():
        if k != "err_flags":
            D[k] = v
    mms, per_col = st.kokkos_evaluate_conservation(D, G.DT, per_column=True)
    assert np.array_equal(per_col, fx["out/evaluate_conservation/diag"], equal_nan=True)
    assert checked >= 130
    D.close()


def _fixture_device_state(module, rows, pft, optics, **scalars):
    d = F.load(module)
    D = st.ELMState(len(rows))
    D.set_pft(pft)
    D.set_snicar(optics)
    D.set_land(**F.TEST_LAND)
    if scalars:
        D.set_scalars(**scalars)
    nlev = {k: v[1] for k, v in D.fields.items()}
    fin, oin = F.split(d, "in/", rows, nlev)
    fout, _ = F.split(d, "out/", rows, nlev)
    for k, v in fin.items():
        D[k] = np.nan_to_num(v, nan=0.0) if D.fields[k][2] != np.float64 else v
    D["vtype"] = np.full(len(rows), F.TEST_LAND["vtype"], np.int32)
    D["veg_active"] = np.ones(len(rows), np.uint8)
    return D, oin, fout


def test_canopy_fluxes_fixture_on_device(min_total=8633, max_loose=120, both_day_and_night=True):
    """The reference's CanopyFluxes fixture (test/test_CanFlux.cc, test/data/CanopyFluxes_{IN,OUT}.txt, 97 steps, 50 by day
    and 47 by night) through the HIP kernels: driven as the reference's test drives the physics - ELM's own forc_rho /
    forc_po2 / forc_pco2 handed in (elmk_canopy_fluxes_given; ELM ran 397.84 ppm CO2, the wrapper hard-wires 355) and the
    step's dayl / max_dayl (a state scalar: the steps are grouped by it).  Same bar as the oracle holds against this
    fixture and as the compiled reference itself shows (73 of 8 633 values beyond 1e-15, worst
    3.5e-10): <= 120 values beyond 1e-15, all < 1e-9; and bit-identical to the oracle on the same steps."""
    from oracle import oracle as O

    pft, optics = synth.load_params()
    d = F.load("CanopyFluxes")
    rows = F.select_steps(d, "CanopyFluxes")
    dayl, mdl = d["in/dayl"][rows, 0], d["in/max_dayl"][rows, 0]
    groups = {}
    for i, key in enumerate(zip(dayl.tolist(), mdl.tolist())):
        groups.setdefault(key, []).append(i)
    assert 1 <= len(groups) <= 8
    total = nloose = 0
    worst = 0.0
    nday = 0
    for (dl, ml), idx in groups.items():
        sub = rows[np.array(idx)]
        D, oin, fout = _fixture_device_state("CanopyFluxes", sub, pft, optics, dayl=dl, max_dayl=ml)
        st.canopy_fluxes_given(D, F.TEST_DTIME, oin["forc_rho"][:, 0], oin["forc_po2"][:, 0], oin["forc_pco2"][:, 0])
        # the oracle on the same columns, same entry
        S = O.OracleState(len(sub))
        S.load_params(pft, optics)
        S.set_scalars(**F.TEST_LAND, dayl=dl, max_dayl=ml)
        fin, _ = F.split(d, "in/", sub, S.nlev)
        F.fill_state(S, fin)
        S["vtype"][:] = F.TEST_LAND["vtype"]
        S.canopy_fluxes_given(F.TEST_DTIME, oin["forc_rho"][:, 0], oin["forc_po2"][:, 0], oin["forc_pco2"][:, 0])
        for name, exp in fout.items():
            got = D[name].reshape(len(sub), -1).astype(np.float64)
            ok = F.almost_equal(got, exp) | np.isnan(exp)
            total += ok.size
            nloose += int((~ok).sum())
            if not ok.all():
                # (relative to the field's own scale: a difference of 4e-16 on an h2ocan of 1e-16 is not an error of 100 %)
                worst = max(worst, float(np.where(ok, 0.0, np.abs(got - exp)).max()) / max(float(np.nanmax(np.abs(exp))), 1e-300))
            ref = np.ascontiguousarray(S[name].reshape(len(sub), -1))
            if ref.dtype.kind == "f":
                assert ((got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))).all(), name
        flags, _ = D.error_summary()
        assert (flags & 0x7FF) == 0
        trips = D.canopy_trip_counts()
        assert trips.min() >= 3 and trips.max() <= 41
        nday += int((d["in/parsun_z"][sub].reshape(len(sub), -1)[:, 0] > 0).sum())
        D.close()
    assert total >= min_total and nloose <= max_loose and worst < 1e-9, (total, nloose, worst)
    assert 0 < nday and (nday < len(rows) or not both_day_and_night)  # the day branch (root finds) and, in the first dump, the night branch ran


def test_second_fixture_dump_on_device():
    """The reference's second ELM dump, test/new_data (tests/golden/newdata/, read by no reference test; snow-free summer
    forcing, 97 steps per module) through the HIP kernels: the five streaming modules against the _OUT records, and
    CanopyFluxes as above (a bounded set beyond 1e-15, all < 1e-9, bit-identical to the oracle).  The oracle is pinned
    against the same vectors on the CPU (tests/test_oracle_golden_newdata.py)."""
    with F.dataset("newdata"):
        test_fixture_steps_on_device()
        test_canopy_fluxes_fixture_on_device(min_total=23000, max_loose=150, both_day_and_night=False)


def test_bareground_fluxes_fixture_on_device():
    """The reference's BareGroundFluxes fixture through the HIP kernels, as test/test_BGFlux.cc drives it: frac_veg_nosno
    hard-wired to 0 (:219: the fixture itself never takes the bare branch) and ELM's own forc_rho handed in
    (elmk_bareground_fluxes_given): every value within the reference's own 1e-15 of the _OUT records."""
    pft, optics = synth.load_params()
    d = F.load("BareGroundFluxes")
    rows = F.select_steps(d, "BareGroundFluxes")
    D, oin, fout = _fixture_device_state("BareGroundFluxes", rows, pft, optics)
    D["frac_veg_nosno"] = np.zeros(len(rows), np.int32)
    st.bareground_fluxes_given(D, oin["forc_rho"][:, 0])
    total = 0
    for name, exp in fout.items():
        if name == "frac_veg_nosno":
            continue
        got = D[name].reshape(len(rows), -1).astype(np.float64)
        ok = F.almost_equal(got, exp) | np.isnan(exp)
        total += ok.size
        assert ok.all(), (name, float(np.where(ok, 0.0, F.rel_err(got, exp)).max()))
    assert total >= 2000
    D.close()


def test_tiling_invariance_at_scale():
    """Size-independent property at a BASELINE-scale N: tiling without perturbation must reproduce the base block's
    results in every tile, bit for bit (columns are independent; no cross-column state)."""
    ft = st.field_table()
    nbase, n = 3008, 1_000_000
    cols, scal, soil = synth.make_state(ft, nbase, tier="B", seed=77)
    Dbase = H.device_state(cols, scal, soil)
    st.timestep7(Dbase, DT)
    big = st.ELMState(n)
    pft, optics = synth.load_params()
    big.set_pft(pft); big.set_snicar(optics); big.set_soilcolor(soil["albsat"], soil["albdry"])
    big.set_land(**synth.TEST_LAND); big.set_scalars(**scal)
    for k, v in cols.items():
        big.upload(k, v, col0=0)
    big.tile_columns(nbase, rules=())
    st.timestep7(big, DT)
    idx = np.arange(n) % nbase
    for name in ("t_veg", "h2osno", "albd", "sabg_lyr", "cgrnd", "btran", "rootr", "snl", "qflx_evap_veg", "t_ref2m"):
        exp = Dbase[name][idx]
        got = big[name]
        assert np.array_equal(got, exp, equal_nan=True), name
    fb, _ = Dbase.error_summary()
    fl, _ = big.error_summary()
    assert fb == fl
    Dbase.close(); big.close()


def test_tiling_invariance_at_the_north_star_size():
    """The same property at the north-star size, 10 M columns (BASELINE configs 3 / 5: 64 GB of device state), through the rest
    of advance() as well: the fused step, the soil-column temperature solve, snow hydrology and surface fluxes on 10 M
    columns must reproduce the base block's results in every tile, bit for bit."""
    ft = st.field_table()
    nbase, n = 6016, 10_000_000
    cols, scal, soil = synth.make_state(ft, nbase, tier="B", seed=78)
    Dbase = H.device_state(cols, scal, soil)
    big = st.ELMState(n)
    pft, optics = synth.load_params()
    big.set_pft(pft); big.set_snicar(optics); big.set_soilcolor(soil["albsat"], soil["albdry"])
    big.set_land(**synth.TEST_LAND); big.set_scalars(**scal)
    big.set_snow_age_tables(synth.snow_age_tables())
    for k, v in cols.items():
        big.upload(k, v, col0=0)
    big.tile_columns(nbase, rules=())
    for D in (Dbase, big):
        st.kokkos_init_timestep(D)
        st.timestep7_fused(D, DT)
        st.kokkos_soil_temperature(D, DT)
        st.kokkos_snow_hydrology(D, DT)
        st.kokkos_surface_fluxes(D, DT)
    idx = np.arange(n) % nbase
    for name in ("t_veg", "h2osno", "cgrnd", "snl", "t_grnd", "xmf", "snow_depth", "eflx_soil_grnd", "qflx_top_soil", "btran"):
        assert np.array_equal(big[name], Dbase[name][idx], equal_nan=True), name
    # level arrays: a window at the far end of the state (the last tiles) instead of 1.6 GB per field
    w0, wn = n - 200_000, 200_000
    for name in ("t_soisno", "h2osoi_ice", "dz", "snw_rds", "mss_bcphi", "imelt", "sabg_lyr"):
        got = big.download(name, col0=w0, n=wn)
        assert np.array_equal(got, Dbase[name][idx[w0:]], equal_nan=True), name
    fb, _ = Dbase.error_summary()
    fl, _ = big.error_summary()
    assert fb == fl and big.device_bytes > 60e9
    Dbase.close(); big.close()


@pytest.mark.parametrize("n", [1, 63, 65, 257, 1000])
def test_ragged_sizes(n):
    """Column counts that are not multiples of the wave (64) or workgroup (256) size: the padded tail of every level
    row must neither be read into results nor written (queue positions, record blocks of 8, work lists)."""
    D, S = _pair(n, "B", 40 + n)
    for step in range(2):  # the second step runs with scheduling hints from the first
        st.timestep7(D, DT)
        S.timestep7(DT)
        assert D.ncols == n and D["t_veg"].shape == (n,)
        _check(D, S, f"n={n} step {step}", bitwise=True)
        for k, v in S.fields.items():  # re-sync (rounding differences are not carried into the next step)
            if k != "err_flags":
                D[k] = v
    D.close()


def _uniform_case(mutate, n=3000, seed=77):
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=seed)
    mutate(cols)
    S = H.oracle_state(cols, scal, soil)
    D = H.device_state(cols, scal, soil)
    return D, S


def test_all_night_all_bare_all_deep_snow():
    """Degenerate populations: every work list but one is empty (queue kernels see counts of 0 and of ncols)."""
    def night(c):
        c["coszen"][:] = -0.3
    def bare(c):
        c["frac_veg_nosno"][:] = 0
    def noon(c):
        c["coszen"][:] = 0.9
    for name, mut in (("night", night), ("bare", bare), ("noon", noon)):
        D, S = _uniform_case(mut)
        st.timestep7(D, DT)
        S.timestep7(DT)
        _check(D, S, f"uniform population: {name}", bitwise=True)
        trips = D.canopy_trip_counts()
        veg = S["frac_veg_nosno"] != 0
        assert ((trips > 0) == veg).all() and trips.max() <= 41
        D.close()


def test_error_flags_match_the_reference_throw_sites():
    """Inputs that make the reference throw/assert: the device raises the same per-column bits as the oracle and
    leaves the other columns' results untouched."""
    ft = st.field_table()
    n = 2048
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=91)
    rng = np.random.default_rng(5)
    bad_rds = rng.choice(n, 200, replace=False)
    cols["snw_rds"][bad_rds] = 5000.0  # beyond the Mie table: snow_snicar_impl.hh:74-78 throws
    bad_hgt = rng.choice(n, 200, replace=False)
    for k in ("forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch"):
        cols[k][bad_hgt] = -50.0  # still below the displacement height after canopy_temperature's "+= z0m + displa":
        #                           canopy_fluxes_impl.hh:178 assert
    S = H.oracle_state(cols, scal, soil)
    D = H.device_state(cols, scal, soil)
    st.timestep7(D, DT)
    S.timestep7(DT)
    fd, fo = D["err_flags"] & 0x7FF, S["err_flags"] & 0x7FF
    assert np.array_equal(fd, fo)
    # the cases above did reach their throw sites (only sunlit snowy / vegetated columns can)
    assert ((fo & (1 << 6)) != 0).sum() >= 20 and ((fo & (1 << 1)) != 0).sum() >= 20
    flags, first = D.error_summary()
    assert flags & 0x7FF == int(np.bitwise_or.reduce(fo)) and first == int(np.nonzero(fo)[0][0])
    _check(D, S, "columns without a fatal flag", skip_cols=fo != 0, bitwise=True)
    D.clear_errors()
    assert D.error_summary()[0] == 0
    D.close()


def test_work_lists_are_left_empty_between_calls():
    """Every compacted kernel drains a device-side work list that the same wrapper filled and leaves it empty for the next
    call (no reset launch in front of it: an empty launch costs 4.5 us).  A list that kept its entries would make the next
    call append behind them - same results, more and more work - so the counters themselves are checked: the bare-ground
    list and the five SNICAR queues read (0 entries, head 0) after every wrapper that uses them, per wrapper and fused, on a
    state that fills all of them."""
    D, S = _pair(20000, "B", 61)
    BG, ALB = 1, slice(2, 8)

    def empty(what):
        c = D.work_list_counters()
        assert not c[BG].any() and not c[ALB].any(), (what, c.tolist())

    st.kokkos_frac_wet(D)
    S.frac_wet()
    for rep in range(2):
        st.kokkos_albedo_snicar(D)
        S.albedo_snicar()
        empty(f"albedo_snicar, call {rep}")
    st.kokkos_canopy_hydrology(D, DT)
    st.kokkos_surface_radiation(D)
    st.kokkos_canopy_temperature(D)
    S.canopy_hydrology(DT)
    S.surface_radiation()
    S.canopy_temperature()
    for rep in range(3):  # the same wrapper again and again: each call starts from an empty list
        st.kokkos_bareground_fluxes(D)
        S.bareground_fluxes()
        empty(f"bareground_fluxes, call {rep}")
    st.kokkos_canopy_fluxes(D, DT)
    S.canopy_fluxes(DT)
    empty("canopy_fluxes")
    _check(D, S, "the step made of those calls", bitwise=True)
    for step in range(2):
        st.timestep7_fused(D, DT)
        S.timestep7(DT)
        empty(f"fused step {step}")
    _check(D, S, "fused steps after it", bitwise=True)
    assert (S["frac_veg_nosno"] == 0).sum() > 3000 and (S["snl"] > 1).sum() > 3000  # (the lists were in use)
    D.close()


def test_results_do_not_depend_on_the_schedule():
    """The canopy_fluxes work queue is ordered by the previous call's trip counts (a hint) and filled in workgroup
    arrival order: results must be bit-identical whatever the hint state and the order."""
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, 30000, tier="B", seed=123)
    D = H.device_state(cols, scal, soil)
    outs = []
    for rep in range(3):  # rep 0: no hints; rep 1: exact hints; rep 2: hints from a different state
        for k, v in cols.items():
            D[k] = v
        if rep == 2:
            D["t_veg"] = cols["t_veg"] + 3.0
            st.timestep7(D, DT)
            for k, v in cols.items():
                D[k] = v
        st.timestep7(D, DT)
        outs.append({k: D[k] for k in ("t_veg", "cgrnd", "h2ocan", "eflx_sh_veg", "t_ref2m", "albd", "flx_absdv", "sabg_lyr")})
    for rep in (1, 2):
        for k, v in outs[0].items():
            assert np.array_equal(v, outs[rep][k], equal_nan=True), (rep, k)
    D.close()


@pytest.mark.parametrize("tier,n,seed", [("A", 4700, 31), ("B", 20000, 32), ("B", 1001, 33)])
def test_soil_temperature_next_row(tier, n, seed):
    """kokkos_soil_temperature (the next call of ELMInterface::advance after the seven): thermal properties, the
    21-row pentadiagonal temperature system, its solve, phase change and ground temperature, on bit-identical inputs
    (the device state is re-synchronised from the oracle after the seven wrappers)."""
    D, S = _pair(n, tier, seed)
    st.timestep7(D, DT)
    S.timestep7(DT)
    for k, v in S.fields.items():
        if k != "err_flags":
            D[k] = v
    st.kokkos_soil_temperature(D, DT)
    S.soil_temperature(DT)
    names = ["t_soisno", "t_h2osfc", "t_grnd", "h2osoi_ice", "h2osoi_liq", "h2osfc", "h2osno", "int_snow", "snow_depth",
             "fact", "sabg_chk", "xmf", "xmf_h2osfc", "qflx_h2osfc_ice", "eflx_h2osfc_snow", "qflx_snofrz", "qflx_snow_melt",
             "qflx_snomelt", "eflx_snomelt", "qflx_snofrz_lyr", "imelt"]
    worst, bad = H.compare_states(D, S, names=names, bitwise=True)
    assert not bad, f"soil_temperature {tier}/{n}: not bit-identical: {bad}"
    # nothing else was touched
    others = [k for k in S.fields if k not in names and k != "err_flags"]
    worst, bad = H.compare_states(D, S, names=others, rel=0.0)
    assert not bad, bad
    if tier == "B":
        im = np.bincount(D["imelt"].ravel(), minlength=3)
        assert im[1] > 0 and im[2] > 0 and set(np.unique(D["snl"])) == {0, 1, 2, 3, 4, 5}
    D.close()


def test_surface_fluxes_and_conservation_diagnostics():
    """kokkos_surface_fluxes and kokkos_evaluate_conservation (the calls after soil_temperature), on inputs
    re-synchronised from the oracle; the device (min, max, sum) of the eight diagnostics against numpy."""
    n = 20000
    D, S = _pair(n, "B", 51)
    st.timestep7(D, DT)
    S.timestep7(DT)
    S.soil_temperature(DT)
    for k, v in S.fields.items():
        if k != "err_flags":
            D[k] = v
    st.kokkos_surface_fluxes(D, DT)
    S.surface_fluxes(DT)
    names = ["eflx_sh_grnd", "qflx_evap_soi", "qflx_ev_snow", "qflx_ev_soil", "qflx_ev_h2osfc", "eflx_soil_grnd", "eflx_sh_tot",
             "qflx_evap_tot", "eflx_lh_tot", "qflx_evap_grnd", "qflx_sub_snow", "qflx_dew_snow", "qflx_dew_grnd",
             "qflx_snwcp_liq", "qflx_snwcp_ice", "eflx_lwrad_out", "eflx_lwrad_net", "soil_e_balance"]
    ponded = S["frac_h2osfc"] != 0  # pow(t_h2osfc_bef, 40) ~ 1e97 there (reference quirk)
    worst, bad = H.compare_states(D, S, names=names, bitwise=True)  # pow(x, 40) included: the device pow is the host's
    assert not bad, f"surface_fluxes: not bit-identical: {bad}"
    others = [k for k in S.fields if k not in names and k != "err_flags"]
    worst, bad = H.compare_states(D, S, names=others, rel=0.0)
    assert not bad, bad
    for k in names:  # the conservation wrapper reads identical bits
        D[k] = S.fields[k]
    mms, cols = st.kokkos_evaluate_conservation(D, DT, per_column=True)
    ref = S.evaluate_conservation(DT)
    e = np.abs(cols - ref) / np.maximum(np.abs(ref), 1e-6)
    assert e[~ponded].max() < 1e-12 and e[ponded][:, [0, 1, 2, 3, 4, 5, 7]].max() < 1e-12
    print("conservation diagnostics bit-identical:", np.array_equal(cols, ref, equal_nan=True))
    assert np.array_equal(mms[:, 0], cols.min(axis=0)) and np.array_equal(mms[:, 1], cols.max(axis=0))
    assert np.allclose(mms[:, 2], cols.sum(axis=0), rtol=1e-12, atol=1e-12 * np.abs(cols).sum(axis=0).max())
    D.close()


def test_init_timestep_column_kernel():
    """The per-column kernel of kokkos_init_timestep: every field bit-identical (the column water mass included: the
    levels are summed in the reference's order)."""
    D, S = _pair(5000, "B", 61)
    st.kokkos_init_timestep(D)
    S.init_timestep()
    worst, bad = H.compare_states(D, S, bitwise=True)
    assert not bad, bad
    D.close()


def test_empty_state():
    """Zero columns (a rank that owns nothing after the block split): every entry is a no-op that succeeds."""
    D = st.ELMState(0)
    st.kokkos_init_timestep(D)
    st.timestep7(D, DT)
    st.kokkos_soil_temperature(D, DT)
    st.kokkos_surface_fluxes(D, DT)
    assert D.error_summary() == (0, -1)
    assert D.download("t_soisno").shape == (0, 20)
    D.close()


def test_graph_replay_is_the_same_step():
    """elmk_set_graph: timestep7 captured once and replayed as a HIP graph (side-stream fork / join included) gives the same
    bits as the launch-by-launch step, step after step, also after a change of dt (re-capture) and when switched off."""
    n = 5000
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=81)
    A = H.device_state(cols, scal, soil)
    B = H.device_state(cols, scal, soil)
    B.set_graph(True)
    for dt in (DT, DT, 900.0, DT):
        st.timestep7(A, dt)
        st.timestep7(B, dt)
        for k in A.fields:
            assert np.array_equal(A[k], B[k], equal_nan=True), (dt, k)
    B.set_graph(False)
    st.timestep7(A, DT)
    st.timestep7(B, DT)
    assert all(np.array_equal(A[k], B[k], equal_nan=True) for k in A.fields)
    A.close()
    B.close()


def _same_bits(A, B, what):
    for k in A.fields:
        a, b = A[k], B[k]
        same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind != "f" else bool(
            ((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())
        assert same, (what, k)


@pytest.mark.parametrize("tier,n,seed", [("A", 4700, 91), ("B", 12000, 92), ("B", 1, 93), ("B", 63, 94), ("B", 257, 95), ("B", 1001, 96)])
def test_fused_timestep_is_the_same_step(tier, n, seed):
    """elmk_timestep7_fused (BASELINE config 5's launch structure: the five streaming wrappers between albedo and the
    leaf-temperature iteration as one pass per column) against elmk_timestep7 on the same state and against the oracle:
    bit-identical in EVERY field (state, error flags, trip counts) over three chained steps, ragged sizes included."""
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=seed)
    U = H.device_state(cols, scal, soil)
    Fz = H.device_state(cols, scal, soil)
    S = H.oracle_state(cols, scal, soil)
    for step in range(3):
        st.timestep7(U, DT)
        st.timestep7_fused(Fz, DT)
        S.timestep7(DT)
        _same_bits(U, Fz, f"fused vs unfused, {tier}/{n}/step {step}")
        _check(Fz, S, f"fused vs oracle, {tier}/{n}/step {step}", bitwise=True)
        assert np.array_equal(U.canopy_trip_counts(), Fz.canopy_trip_counts())
    U.close()
    Fz.close()


@pytest.mark.parametrize("tier,n,seed", [("B", 300_000, 98), ("A", 262_144, 99)])
def test_fused_timestep_large_launch_structure(tier, n, seed):
    """The launch structure elmk_timestep7_fused only takes at >= 262 144 columns - the single-layer SNICAR queue dealt between
    k_fz_pre's tiles (k_fz_snicar_pre), the merged deep SNICAR queues, the bare-ground list kernel on a side stream beside
    k_cf_iterate / k_cf_finish - against elmk_timestep7 and against the oracle: bit-identical in EVERY field (cgrnd / cgrnds /
    cgrndl, which k_bg_flux and k_cf_finish would both store on a bare column, rootr, btran, tssbef, eff_porosity, the SNICAR
    products included) over two chained steps.  This is the structure the benchmark runs."""
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=seed)
    U = H.device_state(cols, scal, soil)
    Fz = H.device_state(cols, scal, soil)
    S = H.oracle_state(cols, scal, soil)
    for step in range(2):
        st.timestep7(U, DT)
        st.timestep7_fused(Fz, DT)
        S.timestep7(DT)
        _same_bits(U, Fz, f"fused vs unfused, {tier}/{n}/step {step}")
        _check(Fz, S, f"fused vs oracle, {tier}/{n}/step {step}", bitwise=True)
        assert np.array_equal(U.canopy_trip_counts(), Fz.canopy_trip_counts())
        for name in ("cgrnd", "cgrnds", "cgrndl"):  # canopy_fluxes' compute_flux leaves them 0 on every non-vegetated column
            assert not Fz[name][S.fields["frac_veg_nosno"] == 0].any(), name
    c = Fz.work_list_counters()
    assert not c[1].any() and not c[2:8].any(), ("a work list was left non-empty", c.tolist())
    U.close()
    Fz.close()


@pytest.mark.parametrize("tier,n,seed", [("B", 20000, 101), ("B", 300_000, 102)])
def test_half_workgroup_iteration_is_the_same_step(tier, n, seed):
    """elmk_set_option(ELMK_OPT_CF_HALF_WORKGROUPS): the leaf-temperature iteration in 256-thread workgroups, one per CU (the launch shape
    for a block of columns that runs beside another context's kernels) is the same kernel body on another workgroup size: fused and per
    wrapper, bit-identical in every field to the product's shape and to the oracle, trip counts included; switching back works."""
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, n, tier=tier, seed=seed)
    U = H.device_state(cols, scal, soil)
    Hf = H.device_state(cols, scal, soil)
    S = H.oracle_state(cols, scal, soil)
    Hf.set_option(st.OPT_CF_HALF_WORKGROUPS, 1)
    for step in range(3):
        st.timestep7_fused(U, DT)
        if step == 1:
            st.timestep7(Hf, DT)          # the per-wrapper canopy_fluxes takes the option too
        else:
            st.timestep7_fused(Hf, DT)
        S.timestep7(DT)
        _same_bits(U, Hf, f"half workgroups, {tier}/{n}/step {step}")
        _check(Hf, S, f"half workgroups vs oracle, {tier}/{n}/step {step}", bitwise=True)
        assert np.array_equal(U.canopy_trip_counts(), Hf.canopy_trip_counts())
        if step == 1:
            Hf.set_option(st.OPT_CF_HALF_WORKGROUPS, 0)
    # an unknown option is refused (and changes nothing); the option also holds under graph replay
    assert Hf.lib.elmk_set_option(Hf.ctx, 99, 1) < 0
    if n <= 20000:
        Hf.set_option(st.OPT_CF_HALF_WORKGROUPS, 1)
        Hf.set_graph(True)
        for step in range(2):
            st.timestep7_fused(U, DT)
            st.timestep7_fused(Hf, DT)
            _same_bits(U, Hf, f"half workgroups under graph replay, step {step}")
    U.close()
    Hf.close()


def test_fused_timestep_other_land_units_and_graph():
    """The fused step on the land units that take the short branches (wetland, land ice, lake, urban), mixed with unfused
    steps on the same context (the two launch structures share the queue scratch), and replayed as a HIP graph."""
    for land in (dict(ltype=6, ctype=0, vtype=0, urbpoi=0, lakpoi=0), dict(ltype=3, ctype=0, vtype=0, urbpoi=0, lakpoi=0),
                 dict(ltype=5, ctype=0, vtype=0, urbpoi=0, lakpoi=1), dict(ltype=7, ctype=71, vtype=0, urbpoi=1, lakpoi=0)):
        D, S = _pair(2000, "B", 21, land)
        st.timestep7_fused(D, DT)
        S.timestep7(DT)
        _check(D, S, f"fused, land {land}", bitwise=True)
        D.close()
    ft = st.field_table()
    cols, scal, soil = synth.make_state(ft, 5000, tier="B", seed=97)
    A = H.device_state(cols, scal, soil)
    B = H.device_state(cols, scal, soil)
    B.set_graph(True)
    for i, dt in enumerate((DT, DT, 900.0, DT)):
        (st.timestep7_fused if i % 2 == 0 else st.timestep7)(A, dt)  # alternate the two launch structures on A
        st.timestep7_fused(B, dt)                                    # fused, replayed as a graph, on B
        _same_bits(A, B, f"mixed / graph step {i}")
    A.close()
    B.close()


def test_get_forcing_and_phenology():
    """The per-column functors kokkos_init_timestep runs first - get_forcing's eight ComputeAtmForcing_* functors (specific-
    and relative-humidity streams) and ComputePhenology - on records / months on both sides of every clamp: every field
    bit-identical to the oracle, which is pinned bit for bit against the reference's own headers (test_oracle_vs_ref)."""
    rng = np.random.default_rng(5)
    for rh in (False, True):
        D, S = _pair(20000, "B", 71)
        if rh:
            q = np.clip(S["atm_qbot"] * 4000.0, 0.0, 100.0)
            S["atm_qbot"][...] = q
            D["atm_qbot"] = q
        e = rng.random(8)
        st.get_forcing(D, 1.0 - e, e, rh)
        S.get_forcing(1.0 - e, e, rh)
        _check(D, S, f"get_forcing rh={rh}", bitwise=True)
        assert (S["forc_tbot"] == 323.0).any() and (S["forc_pbot"] == 4.0e4).any() and (S["forc_lwrad"] != S["atm_flds"][:, 0]).any()
        D.close()
    D, S = _pair(20000, "B", 72)
    vt = S["vtype"].copy()
    vt[::7] = 0
    vt[1::7] = 14
    S["vtype"][...] = vt
    D["vtype"] = vt
    st.compute_phenology(D, 0.3, 0.7)
    S.phenology(0.3, 0.7)
    _check(D, S, "phenology", bitwise=True)
    # and into the step: phenology -> forcing -> init_timestep kernel -> the seven wrappers, still bit-identical
    e = rng.random(8)
    st.get_forcing(D, 1.0 - e, e)
    S.get_forcing(1.0 - e, e)
    st.kokkos_init_timestep(D)
    S.init_timestep()
    st.timestep7(D, DT)
    S.timestep7(DT)
    _check(D, S, "init_timestep functors -> timestep7", bitwise=True)
    D.close()


def _libm():
    import ctypes
    import ctypes.util

    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for f in ("exp", "log", "log10", "atan", "tanh", "cos", "erf", "acos", "expm1"):
        getattr(m, f).restype = ctypes.c_double
        getattr(m, f).argtypes = [ctypes.c_double]
    m.pow.restype = ctypes.c_double
    m.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    return m


def test_device_math_bits():
    """Every <cmath> function of the hot path - exp, log, log10, pow, atan, tanh, cos, erf, acos (+ expm1 under tanh) - on
    the device returns the BITS of the host libm the oracle (and the reference) calls (elmkernels_amd/csrc/elmk_math.h
    restates glibc's algorithms).  Arguments: the ranges the physics uses, the whole exponent range, random bit
    patterns, specials.  (cos is restated for |x| < 105414350, see the header.)"""
    from tests import _parity_mode

    if not _parity_mode.BITWISE_VALID:
        pytest.skip("the host libm is not the one elmk_math.h restates (tests/test_math_host.py is the guard for that)")
    m = _libm()
    rng = np.random.default_rng(20261004)
    n = 200_000
    D = st.ELMState(64)
    spec = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                     0.5, 2.0, 3.0, 4.0, 1e-300, 709.78, 709.79, -745.13, -745.14, -708.4, 1023.9, 1024.0, 0.0625, 16.0])
    bits = rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64)
    xs = {
        "exp": np.concatenate([spec, (rng.random(n) - 0.5) * 1500, (rng.random(n) - 0.5) * 2, bits, (rng.random(n) - 0.5) * 80]),
        "log": np.concatenate([spec, np.exp((rng.random(n) - 0.5) * 1400), 0.9 + 0.2 * rng.random(n), np.abs(bits), 400 * rng.random(n)]),
        "atan": np.concatenate([spec, (rng.random(n) - 0.5) * 40, (rng.random(n) - 0.5) * 2, bits, np.exp((rng.random(n) - 0.5) * 100)]),
    }
    xs["log10"] = xs["log"]
    xs["tanh"] = xs["expm1"] = xs["erf"] = np.concatenate([xs["exp"], (rng.random(n) - 0.5) * 12])
    xs["cos"] = np.concatenate([spec[np.abs(spec) < 1e8], np.pi * rng.random(n), (rng.random(n) - 0.5) * 2e8, (rng.random(n) - 0.5) * 20,
                                np.where(np.abs(bits) < 1e8, bits, 0.5)])
    xs["acos"] = np.concatenate([spec, 2 * rng.random(n) - 1, np.sign(rng.random(n) - 0.5) * (1 - 0.04 * rng.random(n) * rng.random(n)), bits])
    for fn, x in xs.items():
        want = np.array([getattr(m, fn)(float(v)) for v in x])
        got = D.math_eval(fn, x)
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), f"{fn}: {np.count_nonzero(~same)} of {x.size} differ, first x = {x[~same][0]!r}"
    ys = np.array([3.0, 4.0, 0.333, -0.333, 0.666666666666, 1.5, 2.0, 0.5, -1.0, 40.0, 0.25, 1 / 3, -0.5, 2.5, 7.0, -2.0])
    px = np.concatenate([np.repeat(spec, spec.size), 1000 * rng.random(n), np.exp((rng.random(n) - 0.5) * 1400), bits, 2 * rng.random(n),
                         -rng.integers(0, 50, n).astype(np.float64)])
    py = np.concatenate([np.tile(spec, spec.size), ys[rng.integers(0, ys.size, n)], (rng.random(n) - 0.5) * 100,
                         rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64), (rng.random(n) - 0.5) * 10,
                         rng.integers(-10, 11, n) * 0.5])
    want = np.array([m.pow(float(a), float(b)) for a, b in zip(px, py)])
    got = D.math_eval("pow", px, py)
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"pow: {np.count_nonzero(~same)} of {px.size} differ, first (x, y) = {(px[~same][0], py[~same][0])!r}"
    D.close()
