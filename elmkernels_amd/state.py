"""ELMState: the host-side mirror of the reference's ELM::ELMState (src/data/elm_state.h:182-225) on top of
the libelmk context, and the seven physics entry points with the reference's wrapper names.

    S = ELMState(ncols, device=0)
    S["t_soisno"] = arr            # [ncols, 20], reference layout ([column][level])
    S.set_land(ltype=1, ctype=1, vtype=12)
    kokkos_canopy_hydrology(S, dt) # == ELM::kokkos_canopy_hydrology(S, dt) of driver/kokkos
    out = S["h2osno"]              # download

Everything numerical happens in HIP kernels behind the C ABI; this module only moves arrays and arguments.
"""
import ctypes as C

import numpy as np

from . import _lib as L

DTYPES = {0: np.float64, 1: np.int32, 2: np.uint8, 3: np.uint32}
LAYOUT_COL_MAJOR, LAYOUT_SOA = 0, 1
OPT_CF_HALF_WORKGROUPS = 1  # elmk_set_option

# member order of ELM::PFTDataPSN (src/data/pft_data.h:20-24)
PSN_FIELDS = (
    "fnr act25 kcha koha cpha vcmaxha jmaxha tpuha lmrha vcmaxhd jmaxhd tpuhd lmrhd lmrse qe theta_cj "
    "bbbopt mbbopt c3psn slatop leafcn flnr fnitr dleaf smpso smpsc tc_stress"
).split()
ALB_SOURCES = ["rholvis", "rholnir", "rhosvis", "rhosnir", "taulvis", "taulnir", "tausvis", "tausnir", "xl"]


def field_table():
    lib = L.load()
    out = {}
    for i in range(lib.elmk_num_fields()):
        nlev, dt = C.c_int(), C.c_int()
        lib.elmk_field_info(i, C.byref(nlev), C.byref(dt))
        out[lib.elmk_field_name(i).decode()] = (i, nlev.value, DTYPES[dt.value])
    return out


def pack_pft(pft):
    """PFTData::get_pft_psn / get_pft_alb (src/data/pft_data_impl.hh:64-116) for all 25 PFTs -> flat tables."""
    psn = np.zeros((25, 27))
    for j, name in enumerate(PSN_FIELDS):
        v = np.asarray(pft[name], dtype=np.float64).reshape(-1)
        psn[:, j] = v[0] if name == "tc_stress" else v[:25]
    alb = np.zeros((25, 9))
    for j, name in enumerate(ALB_SOURCES):
        alb[:, j] = np.asarray(pft[name], dtype=np.float64).reshape(-1)[:25]
    z0mr = np.ascontiguousarray(np.asarray(pft["z0mr"], dtype=np.float64).reshape(-1)[:25])
    displar = np.ascontiguousarray(np.asarray(pft["displar"], dtype=np.float64).reshape(-1)[:25])
    return psn, alb, z0mr, displar


class ELMState:
    def __init__(self, ncols, device=0, lib_path=None):
        self.lib = L.load(lib_path)
        self.ncols = int(ncols)
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.elmk_create(self.ncols, self.device, C.byref(h))
        if rc != 0:
            raise L.ElmkError(f"elmk_create failed ({rc}): {self.lib.elmk_last_error(None).decode()}")
        self.ctx = h
        self.fields = field_table()
        self.scalars = dict(dewmx=0.1, oldfflag=1, dayl=0.0, max_dayl=0.0)
        self.land = dict(ltype=1, ctype=0, vtype=2, urbpoi=0, lakpoi=0)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.elmk_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise L.ElmkError(f"{what} failed ({rc}): {self.lib.elmk_last_error(self.ctx).decode()}")
        return rc

    # -- arrays -----------------------------------------------------------------------------------
    def upload(self, name, arr, col0=0, layout=LAYOUT_COL_MAJOR):
        fid, nlev, dt = self.fields[name]
        a = np.ascontiguousarray(arr, dtype=dt)
        n = a.size // nlev
        if a.size != n * nlev:
            raise ValueError(f"{name}: size {a.size} is not a multiple of nlev {nlev}")
        self._chk(self.lib.elmk_upload(self.ctx, fid, a.ctypes.data, col0, n, layout), f"upload({name})")

    def download(self, name, col0=0, n=None, layout=LAYOUT_COL_MAJOR, out=None):
        fid, nlev, dt = self.fields[name]
        n = self.ncols - col0 if n is None else n
        shape = (n,) if nlev == 1 else ((n, nlev) if layout == LAYOUT_COL_MAJOR else (nlev, n))
        if out is None:
            out = np.empty(shape, dtype=dt)
        assert out.shape == shape and out.dtype == dt and out.flags.c_contiguous
        self._chk(self.lib.elmk_download(self.ctx, fid, out.ctypes.data, col0, n, layout), f"download({name})")
        return out

    def __setitem__(self, name, arr):
        self.upload(name, arr)

    def __getitem__(self, name):
        return self.download(name)

    def fill(self, name, value):
        self._chk(self.lib.elmk_fill(self.ctx, self.fields[name][0], float(value)), f"fill({name})")

    def device_ptr(self, name):
        return self.lib.elmk_device_ptr(self.ctx, self.fields[name][0])

    @property
    def level_stride(self):
        return self.lib.elmk_level_stride(self.ctx)

    @property
    def device_bytes(self):
        return self.lib.elmk_device_bytes(self.ctx)

    def tile_columns(self, nbase, seed=0x5EEDE1A0, rules=()):
        """Replicate columns [0, nbase) over the rest of the state; rules = [(field, mode, amp)]."""
        arr = (L.Perturb * max(len(rules), 1))()
        for i, (name, mode, amp) in enumerate(rules):
            arr[i] = L.Perturb(self.fields[name][0], int(mode), float(amp))
        self._chk(
            self.lib.elmk_tile_columns(self.ctx, int(nbase), int(seed), len(rules), C.cast(arr, C.c_void_p)),
            "tile_columns",
        )

    def snapshot_fields(self, names):
        ids = (C.c_int * len(names))(*[self.fields[n][0] for n in names])
        self._chk(self.lib.elmk_snapshot_fields(self.ctx, ids, len(names)), "snapshot_fields")

    def restore_fields(self):
        self._chk(self.lib.elmk_restore_fields(self.ctx), "restore_fields")

    # -- parameters -------------------------------------------------------------------------------
    def set_land(self, **kw):
        self.land.update(kw)
        l = self.land
        self._chk(
            self.lib.elmk_set_land(self.ctx, int(l["ltype"]), int(l["ctype"]), int(l["vtype"]), int(l["urbpoi"]), int(l["lakpoi"])),
            "set_land",
        )

    def set_scalars(self, **kw):
        self.scalars.update(kw)
        s = self.scalars
        self._chk(
            self.lib.elmk_set_scalars(self.ctx, float(s["dewmx"]), int(s["oldfflag"]), float(s["dayl"]), float(s["max_dayl"])),
            "set_scalars",
        )

    def set_pft(self, pft):
        psn, alb, z0mr, displar = pack_pft(pft)
        self.set_pft_tables(psn, alb, z0mr, displar)

    def set_pft_tables(self, psn, alb, z0mr, displar):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (psn, alb, z0mr, displar)]
        assert a[0].shape == (25, 27) and a[1].shape == (25, 9) and a[2].shape == (25,) and a[3].shape == (25,)
        self._chk(self.lib.elmk_set_pft(self.ctx, *[x.ctypes.data for x in a]), "set_pft")

    def set_soilcolor(self, albsat, albdry):
        a = np.ascontiguousarray(albsat, dtype=np.float64)
        b = np.ascontiguousarray(albdry, dtype=np.float64)
        assert a.shape == (20, 2) and b.shape == (20, 2)
        self._chk(self.lib.elmk_set_soilcolor(self.ctx, a.ctypes.data, b.ctypes.data), "set_soilcolor")

    def set_snicar(self, tables):
        keep = []
        t = L.SnicarTables()
        for name in L.SNICAR_NAMES:
            a = np.ascontiguousarray(np.asarray(tables[name], dtype=np.float64).reshape(-1))
            if a.size != L.SNICAR_SIZES[name]:
                raise ValueError(f"snicar table {name}: {a.size} values, expected {L.SNICAR_SIZES[name]}")
            keep.append(a)
            setattr(t, name, a.ctypes.data)
        self._chk(self.lib.elmk_set_snicar(self.ctx, C.byref(t)), "set_snicar")

    def set_init_params(self, organic_max, roota_par, rootb_par):
        """organic_max of the parameter file (initialize_elm_kokkos.cc:312) and PFTData::roota_par / rootb_par [25]."""
        a = np.ascontiguousarray(roota_par, dtype=np.float64)
        b = np.ascontiguousarray(rootb_par, dtype=np.float64)
        assert a.shape == (25,) and b.shape == (25,)
        self._chk(self.lib.elmk_set_init_params(self.ctx, float(organic_max), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "set_init_params")

    def set_snow_age_tables(self, tables):
        """SnwRdsTable: tables [3, 11, 31, 8] = snowage_tau, snowage_kappa, snowage_drdt0."""
        t = np.ascontiguousarray(tables, dtype=np.float64)
        assert t.shape == (3, 11, 31, 8)
        self._chk(self.lib.elmk_set_snow_age_tables(self.ctx, *[t[k].ctypes.data_as(C.c_void_p) for k in range(3)]), "set_snow_age_tables")

    # -- control ----------------------------------------------------------------------------------
    def sync(self):
        self._chk(self.lib.elmk_sync(self.ctx), "sync")

    def set_stream(self, hip_stream_handle):
        self._chk(self.lib.elmk_set_stream(self.ctx, C.c_void_p(hip_stream_handle or 0)), "set_stream")

    def error_summary(self):
        flags, first = C.c_uint32(), C.c_int64()
        self._chk(self.lib.elmk_error_summary(self.ctx, C.byref(flags), C.byref(first)), "error_summary")
        return flags.value, first.value

    def clear_errors(self):
        self._chk(self.lib.elmk_clear_errors(self.ctx), "clear_errors")

    def profile_timestep7(self, dt, nsteps):
        ms = (C.c_float * 7)()
        tot = C.c_float()
        self._chk(self.lib.elmk_profile_timestep7(self.ctx, float(dt), int(nsteps), ms, C.byref(tot)), "profile_timestep7")
        return list(ms), tot.value

    def profile_timestep7_fused(self, dt, nsteps):
        ms = (C.c_float * len(KERNEL_NAMES_FUSED))()
        tot = C.c_float()
        self._chk(self.lib.elmk_profile_timestep7_fused(self.ctx, float(dt), int(nsteps), ms, C.byref(tot)), "profile_timestep7_fused")
        return list(ms), tot.value

    def profile_steps(self, dt, nsteps, fused=False):
        """Device time (ms) of each of nsteps steps (HIP events around every step; snapshot restored before each)."""
        ms = (C.c_float * int(nsteps))()
        self._chk(self.lib.elmk_profile_steps(self.ctx, int(bool(fused)), float(dt), int(nsteps), ms), "profile_steps")
        return list(ms)

    def canopy_trip_counts(self):
        """Trips of the leaf-temperature iteration per column in the last canopy_fluxes call (0: not vegetated)."""
        out = np.zeros(self.ncols, dtype=np.int32)
        self._chk(self.lib.elmk_read_scratch(self.ctx, 0, out.ctypes.data_as(C.c_void_p), 0, self.ncols), "read_scratch")
        return out

    def canopy_schedule_hints(self):
        """The scheduler's hint per column: slowly decaying maximum of the trip count (development diagnostics)."""
        out = np.zeros(self.ncols, dtype=np.int32)
        self._chk(self.lib.elmk_read_scratch(self.ctx, 2, out.ctypes.data_as(C.c_void_p), 0, self.ncols), "read_scratch")
        return out

    def work_list_counters(self):
        """(entries, queue head) of every internal work list [nlists, 2]: all zero between two wrapper calls."""
        n = 8
        out = np.zeros(2 * n, dtype=np.uint32)
        self._chk(self.lib.elmk_read_scratch(self.ctx, 3, out.ctypes.data_as(C.c_void_p), 0, 2 * n), "read_scratch")
        return out.reshape(n, 2)

    def read_work(self, offset, count):
        out = np.zeros(int(count), dtype=np.float64)
        self._chk(self.lib.elmk_read_scratch(self.ctx, 1, out.ctypes.data_as(C.c_void_p), int(offset), int(count)), "read_scratch")
        return out

    def profile_wrapper(self, wrapper, dt, nsteps=5):
        """Mean device time (ms, HIP events on the launch stream) of one wrapper; wrapper: index into WRAPPER_NAMES."""
        ms = C.c_float()
        self._chk(self.lib.elmk_profile_wrapper(self.ctx, int(wrapper), float(dt), int(nsteps), C.byref(ms)), "profile_wrapper")
        return ms.value

    def set_option(self, option, value):
        """Launch options (elmk_set_option); OPT_CF_HALF_WORKGROUPS: the leaf-temperature iteration in 256-thread workgroups, one per CU."""
        self._chk(self.lib.elmk_set_option(self.ctx, int(option), int(value)), "set_option")

    def set_graph(self, on=True):
        """timestep7 as one replayed HIP graph (elmk_set_graph)."""
        self._chk(self.lib.elmk_set_graph(self.ctx, int(bool(on))), "set_graph")

    def copy_bandwidth(self, nbytes=1 << 30, iters=20, shape=0):
        """device-to-device copy rate in GB/s (read + write bytes); shape: see elmk_copy_bandwidth_shape"""
        g = C.c_double()
        self._chk(self.lib.elmk_copy_bandwidth_shape(self.ctx, int(nbytes), int(iters), int(shape), C.byref(g)), "copy_bandwidth")
        return g.value


    def math_eval(self, fn, x, y=None):
        """elmk_math.h on the device: fn in MATH_FNS; returns fn(x), x / y or pow(x, y)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            assert y.shape == x.shape
            yp = y.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.elmk_math_eval(self.ctx, MATH_FNS.index(fn), x.ctypes.data_as(C.c_void_p), yp,
                                          out.ctypes.data_as(C.c_void_p), x.size), "math_eval")
        return out


MATH_FNS = ["exp", "log", "log10", "atan", "sqrt", "tanh", "cos", "erf", "acos", "expm1", "div", "pow"]
WRAPPER_NAMES = ["frac_wet", "albedo_snicar", "canopy_hydrology", "surface_radiation", "canopy_temperature",
                 "bareground_fluxes", "canopy_fluxes", "soil_temperature", "surface_fluxes", "snow_hydrology", "advance_physics"]
KERNEL_NAMES = [
    "frac_wet", "albedo_snicar", "canopy_hydrology", "surface_radiation", "canopy_temperature",
    "bareground_fluxes", "canopy_fluxes",
]
# launch groups of elmk_timestep7_fused (include/elmk.h: elmk_profile_timestep7_fused)
KERNEL_NAMES_FUSED = ["prep_frac_wet", "albedo_snicar", "fused_stream", "bareground_list", "canopy_iterate"]


# ---- the L3 wrappers, named as in driver/kokkos/*_kokkos.hh ------------------------------------------
def kokkos_frac_wet(S):
    S._chk(S.lib.elmk_frac_wet(S.ctx), "frac_wet")


def kokkos_albedo_snicar(S):
    S._chk(S.lib.elmk_albedo_snicar(S.ctx), "albedo_snicar")


def kokkos_canopy_hydrology(S, dt):
    S._chk(S.lib.elmk_canopy_hydrology(S.ctx, float(dt)), "canopy_hydrology")


def kokkos_surface_radiation(S):
    S._chk(S.lib.elmk_surface_radiation(S.ctx), "surface_radiation")


def kokkos_canopy_temperature(S):
    S._chk(S.lib.elmk_canopy_temperature(S.ctx), "canopy_temperature")


def kokkos_bareground_fluxes(S):
    S._chk(S.lib.elmk_bareground_fluxes(S.ctx), "bareground_fluxes")


def kokkos_canopy_fluxes(S, dt):
    S._chk(S.lib.elmk_canopy_fluxes(S.ctx, float(dt)), "canopy_fluxes")


def _opt(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def canopy_fluxes_given(S, dt, forc_rho=None, forc_po2=None, forc_pco2=None):
    """L2-level canopy_fluxes: forcing-derived scalars handed in (test/test_CanFlux.cc) instead of derived by the wrapper."""
    a = [_opt(x) for x in (forc_rho, forc_po2, forc_pco2)]
    assert all(x is None or x.shape == (S.ncols,) for x in a)
    S._chk(S.lib.elmk_canopy_fluxes_given(S.ctx, float(dt), *[None if x is None else x.ctypes.data_as(C.c_void_p) for x in a]),
           "canopy_fluxes_given")


def bareground_fluxes_given(S, forc_rho):
    """L2-level bareground_fluxes with ELM's own air density (test/test_BGFlux.cc)."""
    a = _opt(forc_rho)
    assert a.shape == (S.ncols,)
    S._chk(S.lib.elmk_bareground_fluxes_given(S.ctx, a.ctypes.data_as(C.c_void_p)), "bareground_fluxes_given")


def kokkos_soil_temperature(S, dt):
    """Next in ELMInterface::advance after the seven (elm_kokkos_interface.cc:310; soil_temperature_kokkos.cc:6-278)."""
    S._chk(S.lib.elmk_soil_temperature(S.ctx, float(dt)), "soil_temperature")


def kokkos_snow_hydrology(S, dt):
    """snow_hydrology_kokkos.cc:23-188 (between soil_temperature and surface_fluxes in ELMInterface::advance, :313)."""
    S._chk(S.lib.elmk_snow_hydrology(S.ctx, float(dt)), "snow_hydrology")


def get_forcing(S, wt1, wt2, qbot_is_rh=False):
    """ELM::get_forcing (atm_forcing_kokkos.cc:47-75); wt1, wt2: [8] weights of TBOT, PBOT, QBOT|RH, FLDS, FSDS, PREC, WIND, ZBOT."""
    w1 = np.ascontiguousarray(wt1, dtype=np.float64)
    w2 = np.ascontiguousarray(wt2, dtype=np.float64)
    assert w1.shape == (8,) and w2.shape == (8,)
    S._chk(S.lib.elmk_get_forcing(S.ctx, w1.ctypes.data_as(C.c_void_p), w2.ctypes.data_as(C.c_void_p), int(bool(qbot_is_rh))), "get_forcing")


def compute_phenology(S, wt1, wt2):
    """ComputePhenology (phenology_physics_impl.hh:22-69) as update_phenology runs it (phenology_kokkos.cc:59-62)."""
    S._chk(S.lib.elmk_phenology(S.ctx, float(wt1), float(wt2)), "phenology")


def forcing_time_weights(days_since_record, forc_dt):
    """AtmDataManager::forcing_time_weights (atm_data_impl.hh:191-199): (wt1, wt2) for a model time `days_since_record`
    days after the forcing record t_idx, records `forc_dt` days apart."""
    e = days_since_record / forc_dt
    assert 0.0 <= e <= 1.0
    return 1.0 - e, e


def initialize_kokkos_elm(S):
    """The per-column init functions of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428) - cold-start state from
    topography, snow depth, soil texture and PFT; the file reads of that function stay with the caller."""
    S._chk(S.lib.elmk_initialize_state(S.ctx), "initialize_state")


def kokkos_init_timestep(S):
    """The per-column kernel of kokkos_init_timestep (init_timestep_kokkos.cc:55-75)."""
    S._chk(S.lib.elmk_init_timestep(S.ctx), "init_timestep")


def kokkos_surface_fluxes(S, dt):
    """surface_fluxes_kokkos.cc:5-107 (follows soil_temperature in ELMInterface::advance)."""
    S._chk(S.lib.elmk_surface_fluxes(S.ctx, float(dt)), "surface_fluxes")


CONSERVATION_NAMES = ("dtend_column_h2o", "errh2o", "errh2osno", "dwb", "errsol", "errlon", "errseb", "netrad")


def kokkos_evaluate_conservation(S, dt, per_column=False):
    """conserved_quantity_kokkos.cc:8-81 -> min_max_sum [8, 3] (rows: CONSERVATION_NAMES) and, if asked, the
    per-column values [ncols, 8]."""
    mms = np.zeros((8, 3))
    cols = np.zeros((8, S.ncols)) if per_column else None
    S._chk(S.lib.elmk_evaluate_conservation(S.ctx, float(dt), mms.ctypes.data_as(C.c_void_p),
                                            cols.ctypes.data_as(C.c_void_p) if per_column else None), "evaluate_conservation")
    return (mms, np.ascontiguousarray(cols.T)) if per_column else mms


def timestep7_fused(S, dt):
    """The same seven calls with the streaming wrappers between albedo and the leaf-temperature iteration fused into one
    pass per column (elmk_timestep7_fused); bit-identical results."""
    S._chk(S.lib.elmk_timestep7_fused(S.ctx, float(dt)), "timestep7_fused")


def advance_physics(S, dt):
    """Every per-column call of ELMInterface::advance after kokkos_init_timestep, in its order
    (elm_kokkos_interface.cc:289-316): the seven wrappers (fused), soil_temperature, snow_hydrology, surface_fluxes - one
    call (elmk_advance_physics), one HIP graph launch per step with set_graph(True).  Same bits as the ten calls."""
    S._chk(S.lib.elmk_advance_physics(S.ctx, float(dt)), "advance_physics")


def timestep7(S, dt):
    """The seven calls of ELMInterface::advance (driver/kokkos/elm_kokkos_interface.cc:289-307), in order."""
    S._chk(S.lib.elmk_timestep7(S.ctx, float(dt)), "timestep7")


class ELMInterface:
    """Python mirror of the reference's driver class ELM::ELMInterface (driver/kokkos/elm_kokkos_interface.hh:11-28,
    elm_kokkos_interface.cc:38-358) above the C ABI, the counterpart of include/elmk_interface.hpp: same member names, same call
    order in advance(), same PrimaryVars members (src/data/elm_state.h:17-48).  File reads and date arithmetic stay with
    the caller, who uploads fields / forcing records and passes the interpolation weights."""

    PRIMARY_VARS = ("snl", "snow_depth", "frac_sno", "int_snow", "snw_rds", "h2osoi_liq", "h2osoi_ice", "h2osoi_vol", "h2ocan",
                    "h2osno", "h2osfc", "t_soisno", "t_grnd", "t_h2osfc", "t_h2osfc_bef", "nrad", "dz", "zsoi", "zisoi")

    def __init__(self, ncols, device=0):
        self.S = ELMState(ncols, device=device)
        self.conservation = None

    def setup(self, land, scalars, pft, snicar, soilcolor, snow_age_tables, init_params=None, graph=True):
        """ELMInterface::setup (elm_kokkos_interface.cc:58-267) minus the file reads."""
        S = self.S
        S.set_land(**land)
        S.set_scalars(**scalars)
        S.set_pft(pft)
        S.set_snicar(snicar)
        S.set_soilcolor(*soilcolor)
        S.set_snow_age_tables(snow_age_tables)
        if init_params is not None:
            S.set_init_params(*init_params)
        S.set_graph(bool(graph))

    def initialize(self):
        """The per-column part of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428), after the uploads."""
        initialize_kokkos_elm(self.S)

    def advance(self, dt_seconds, forc_wt1, forc_wt2, month_wt1, month_wt2, qbot_is_rh=False):
        """ELMInterface::advance (elm_kokkos_interface.cc:269-322); returns False like the reference."""
        S = self.S
        compute_phenology(S, month_wt1, month_wt2)
        get_forcing(S, forc_wt1, forc_wt2, qbot_is_rh)
        kokkos_init_timestep(S)
        advance_physics(S, dt_seconds)
        self.conservation = kokkos_evaluate_conservation(S, dt_seconds)
        flags, col = S.error_summary()
        if flags & 0xC7FF:  # ELMK_ERR_FATAL_MASK
            raise RuntimeError(f"ELM physics error flags {flags:#x}, first at column {col}")
        return False

    def getPrimaryVars(self):
        """ELMInterface::getPrimaryVars / copyPrimaryVars (:324-356): the PrimaryVars members as host arrays."""
        return {k: self.S[k] for k in self.PRIMARY_VARS}

    def close(self):
        self.S.close()
