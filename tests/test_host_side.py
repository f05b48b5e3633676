"""CPU-only checks of everything around the HIP kernels: the C-ABI library loads and exports every symbol that
include/elmk.h declares, schema consistency between the product and the oracle, the column decomposition, the
synthetic-state generator, and the fixture converter.  No compute call is made (there is no GPU here, and the
product has no CPU path)."""
import ctypes
import os
import re

import numpy as np
import pytest

import elmkernels_amd as E
from elmkernels_amd import _lib, decomp, synth
from elmkernels_amd import state as st
from oracle import oracle as O
from tests import fixtures as F
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "elmk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(elmk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"libelmk.so lacks {name} declared in include/elmk.h"
    # and the ctypes table covers the whole header (no entry point is bound by guesswork)
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)


def test_fp32_state_build_has_the_same_abi():
    """libelmk_f32.so (BASELINE config 5's report-only variant: fp64 fields stored as fp32) is the same ABI, symbol for symbol,
    says what it stores, and refuses to compute without a device like the product."""
    lib = _lib.load()
    lib32 = _lib.load(_lib.F32_LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib32, name), f"libelmk_f32.so lacks {name}"
    assert lib.elmk_state_real_bytes() == 8 and lib32.elmk_state_real_bytes() == 4
    assert lib32.elmk_num_fields() == lib.elmk_num_fields()
    import torch

    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        assert lib32.elmk_create(16, 0, ctypes.byref(h)) == -2 and not h.value


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product refuses to create a context (it must never compute on the CPU)."""
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.elmk_create(16, 0, ctypes.byref(h))
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in lib.elmk_last_error(None)
    with pytest.raises(_lib.ElmkError):
        E.ELMState(16)


def test_field_schema_matches_reference_state_and_oracle():
    ft = st.field_table()
    oracle_ft = H.field_table_from_oracle()
    assert set(ft) - {"err_flags"} == set(oracle_ft)
    for name, (_, nlev, dt) in oracle_ft.items():
        assert ft[name][1] == nlev and np.dtype(ft[name][2]) == np.dtype(dt), name
    # extents of ELMStateViews (src/data/elm_state_impl.hh:48-364)
    assert ft["t_soisno"][1] == 20 and ft["zisoi"][1] == 21 and ft["sabg_lyr"][1] == 6 and ft["watsat"][1] == 15
    assert ft["snw_rds"][1] == 5 and ft["albd"][1] == 2 and ft["veg_active"][2] == np.uint8 and ft["snl"][2] == np.int32
    lib = _lib.load()
    assert lib.elmk_field_id(b"t_soisno") == ft["t_soisno"][0] and lib.elmk_field_id(b"nope") == -1
    assert lib.elmk_field_name(ft["cgrnd"][0]) == b"cgrnd"


def test_block_decomposition_matches_reference_rule():
    """create_domain_decomposition_1D (src/utils/utils.cc:27-44): first N % P ranks own one extra column."""
    for n, p in ((10, 3), (80_000_000, 8), (7, 8), (0, 4), (1_000_001, 2)):
        r = decomp.all_ranges(n, p)
        assert sum(c for _, c in r) == n
        assert r[0][0] == 0 and all(r[i][0] + r[i][1] == r[i + 1][0] for i in range(p - 1))
        sizes = [c for _, c in r]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert decomp.block_range(10, 3, 0) == (0, 4) and decomp.block_range(10, 3, 2) == (7, 3)
    with pytest.raises(ValueError):
        decomp.block_range(10, 0, 0)


def test_synthetic_states_are_deterministic_and_cover_the_branches():
    ft = H.field_table_from_oracle()
    a, sa, _ = synth.make_state(ft, 6016, tier="B", seed=5)
    b, sb, _ = synth.make_state(ft, 6016, tier="B", seed=5)
    assert sa == sb and all(np.array_equal(a[k], b[k]) for k in a)
    c, _, _ = synth.make_state(ft, 6016, tier="B", seed=6)
    assert not np.array_equal(a["snl"], c["snl"])
    assert set(np.unique(a["snl"])) == {0, 1, 2, 3, 4, 5} and set(np.unique(a["vtype"])) == {12, 14}
    assert a["do_capsnow"].any() and (a["h2osno"][a["do_capsnow"] == 1] > 1000).all()
    tA, _, _ = synth.make_state(ft, 470, tier="A", seed=1)
    assert (tA["snl"] == 0).all() and (tA["frac_veg_nosno"] == 1).all()
    # a snow layer's water content is consistent with h2osno
    lay = a["snl"] > 0
    tot = (a["h2osoi_ice"][:, :5] + a["h2osoi_liq"][:, :5]).sum(axis=1)
    assert np.allclose(tot[lay], a["h2osno"][lay])


def test_pft_and_snicar_packing():
    pft, optics = synth.load_params()
    psn, alb, z0mr, displar = st.pack_pft(pft)
    S = O.OracleState(1)
    S.load_params(pft, optics)
    assert np.array_equal(psn, S.pft_psn) and np.array_equal(alb, S.pft_alb)
    assert np.array_equal(z0mr, S.z0mr) and np.array_equal(displar, S.displar)
    assert psn[12, st.PSN_FIELDS.index("c3psn")] == 1.0 and psn[14, st.PSN_FIELDS.index("c3psn")] == 0.0
    assert sum(_lib.SNICAR_SIZES.values()) == 18 * 5 + 6 * 5 * 1471 + 6 * 50 + 400
    for name in _lib.SNICAR_NAMES:
        assert optics[name].size == _lib.SNICAR_SIZES[name], name


def test_golden_fixture_shapes():
    for module, nsteps in (("CanopyHydrology", 49), ("CanopyFluxes", 97), ("SurfaceAlbedo", 95)):
        d = F.load(module)
        assert len(d["steps"]) == nsteps
    d = F.load("CanopyHydrology")
    assert d["in/t_soisno"].shape == (49, 20) and d["in/zi"].shape == (49, 21)
    assert np.isnan(d["in/qflx_snwcp_ice"]).any() and (d["in/qflx_irrig"] == 1e36).any()  # sentinels kept as data


def test_header_is_plain_c99():
    """INTEGRATION.md promises include/elmk.h is plain C: a C99 consumer compiles with -pedantic -Werror (syntax only:
    nothing is linked or run here)."""
    import subprocess

    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_consumer.c")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    # and the consumer really exercises the header: every physics entry point it declares appears in the C file
    src = open(os.path.join(ROOT, "tests", "c", "abi_consumer.c")).read()
    for name in ("elmk_create", "elmk_upload", "elmk_timestep7", "elmk_soil_temperature", "elmk_error_summary", "elmk_destroy"):
        assert name in src


def test_cpp_interface_mirror_compiles_and_links():
    """include/elmk_interface.hpp (the C++ mirror of ELM::ELMInterface above the C ABI) and the driver loop that uses it
    (examples/elm_interface_demo.cc) compile with -Wall -Wextra -Werror and link against libelmk; the class has the
    reference's member functions (elm_kokkos_interface.hh:16-20).  Nothing is run here (no GPU)."""
    import subprocess
    import tempfile

    from elmkernels_amd import _lib as L

    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "elm_interface_demo.cc"), "-L" + os.path.dirname(L.LIB_PATH), "-lelmk",
                            "-o", os.path.join(d, "demo")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
    hdr = open(os.path.join(ROOT, "include", "elmk_interface.hpp")).read()
    for member in ("void setup(", "bool advance(", "void copyPrimaryVars(", "getPrimaryVars()"):
        assert member in hdr, member


def test_solar_geometry_matches_the_reference_sources():
    """The host scalars kokkos_init_timestep computes before its kernels - step-averaged cos(zenith), day length, maximum day
    length (init_timestep_kokkos.cc:26-34) - as include/elmk_interface.hpp restates them, against the reference's own
    incident_shortwave.cc and day_length.cc compiled into oracle/_ref: bit for bit over latitudes pole to pole (the poles
    themselves included), every longitude, every day of the year, steps from a minute to a day."""
    import ctypes as C
    import subprocess
    import tempfile

    from oracle import oracle as O

    if not O.have_ref() or not hasattr(O.Reference().R, "elmref_solar"):
        pytest.skip("oracle/_ref/libelmref.so not built here (or predates elmref_solar)")
    R = O.Reference().R
    rng = np.random.default_rng(17)
    n = 200_000
    lat = (rng.random(n) - 0.5) * np.pi
    lat[:4] = [np.pi / 2, -np.pi / 2, 0.0, 1.2]
    lon = (rng.random(n) - 0.5) * 4 * np.pi
    dt = rng.choice([60.0, 1800.0, 3600.0, 10800.0, 86400.0], n)
    jday = 1.0 + 365.0 * rng.random(n)
    jday[::7] = np.floor(jday[::7])  # midnight: the fractional day is exactly zero
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "solar.so")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "c", "solar_shim.cc"), "-o", so])
        L = C.CDLL(so)
        out = {}
        for name, fn in (("mine", L.elmk_test_solar), ("ref", R.elmref_solar)):
            fn.argtypes = [C.c_int64] + [C.c_void_p] * 7
            fn.restype = None
            o = [np.zeros(n) for _ in range(3)]
            fn(n, lat.ctypes.data, lon.ctypes.data, dt.ctypes.data, jday.ctypes.data, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data)
            out[name] = o
    for a, b, what in zip(out["mine"], out["ref"], ("average_cosz", "daylength", "max_daylength")):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), what
    assert (out["ref"][0] > 0).mean() > 0.3 and (out["ref"][0] == 0).mean() > 0.2  # day and night both sampled


def test_parity_bar_is_announced():
    """The bar the device-vs-oracle tests hold is decided once per session (tests/conftest.py) and printed in the report header;
    it must be one of the two documented ones."""
    from tests import _parity_mode as M

    assert isinstance(M.BITWISE_VALID, bool) and M.REASON


def test_bench_cpu_budget_and_floor(tmp_path, monkeypatch):
    """bench.py's host logic that needs no GPU: the CPU budget honours the affinity mask and the cgroup quota (128 OpenMP
    threads inside a 16-CPU quota were the 'CPU baseline'), and roofline.floor is formed from the committed counter tables only when
    they belong to this build and this column count."""
    import json

    import bench

    cap, info = bench.HOST_CPU_BUDGET
    assert 1 <= cap <= info["affinity"] and (info["cgroup_quota_cpus"] is None or cap <= max(1, round(info["cgroup_quota_cpus"])))
    # a traffic table and a compute table of "this build": two kernels, one bandwidth-bound, one arithmetic-bound
    h = bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    tag = bench.PROFILE_TAG
    json.dump({"source_hash": h, "columns": 1_000_000, "kernels": {"elmk::k_a": {"hbm_bytes_per_launch": 5.0e9}, "elmk::k_b": {"hbm_bytes_per_launch": 1.0e9},
                                                                      "elmk::k_copy": {"hbm_bytes_per_launch": 2.0e9}}},
              open(prof / f"{tag}_hbm_traffic_pmc_tierA.json", "w"))
    k = dict(SQ_INSTS_VALU_MUL_F64=0.0, SQ_INSTS_VALU_ADD_F64=0.0, SQ_INSTS_VALU_TRANS_F64=0.0)
    json.dump({"source_hash": h, "kernels": {"k_a": dict(k, SQ_INSTS_VALU=1.0e6, SQ_INSTS_VALU_FMA_F64=1.0e6),
                                             "k_b": dict(k, SQ_INSTS_VALU=4.0e8, SQ_INSTS_VALU_FMA_F64=4.0e8)}},
              open(prof / f"{tag}_compute_pmc_tierA.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    fl = bench.step_floor("A", 1_000_000, 2.5, 5000.0)
    assert fl is not None
    bytes_ms = 6.0e9 / 5000.0e9 * 1e3                      # 1.2 ms: the calibration copy is not part of the step
    valu_ms = (1.0e6 + 4.0e8) * 2.2 / 1024 * 1e-6          # 0.86 ms
    assert abs(fl["bytes_ms"] - bytes_ms) < 1e-3 and abs(fl["valu_ms"] - valu_ms) < 1e-3
    assert abs(fl["floor_ms"] - max(bytes_ms, valu_ms)) < 1e-3 and abs(fl["frac_of_floor"] - fl["floor_ms"] / 2.5) < 1e-3
    assert abs(fl["serial_floor_ms"] - (1.0 + 4.0e8 * 2.2 / 1024 * 1e-6)) < 1e-3   # k_a at its byte roof + k_b at its VALU roof
    assert bench.step_floor("A", 10_000_000, 2.5, 5000.0) is None                  # another column count: no claim
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "another build")
    assert bench.step_floor("A", 1_000_000, 2.5, 5000.0) is None                   # another build: no claim
