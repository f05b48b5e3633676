"""ctypes front-end for the parity oracle (TEST INFRASTRUCTURE - see oracle/elm_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

  Oracle            libelmoracle.so - the plain-C restatement (OpenMP over columns)
  Reference         _ref/libelmref.so - the reference's own headers compiled by oracle/Makefile
                    (present only where it was built in the build container; None otherwise)
  OracleState       the [col][lev] state container both of them operate on
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")

KIND_DTYPE = {0: np.float64, 1: np.int32, 2: np.uint8}

PSN_FIELDS = (
    "fnr act25 kcha koha cpha vcmaxha jmaxha tpuha lmrha vcmaxhd jmaxhd tpuhd lmrhd lmrse qe theta_cj "
    "bbbopt mbbopt c3psn slatop leafcn flnr fnitr dleaf smpso smpsc tc_stress"
).split()

SNICAR_TABLES = (
    [(f"{p}_{s}", 5) for s in ("oc1", "oc2", "dst1", "dst2", "dst3", "dst4") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + [(f"{p}_snw_{s}", 5 * 1471) for s in ("drc", "dfs") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + [(f"{p}_{s}", 50) for s in ("bc1", "bc2") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + [("bcenh", 400)]
)


def build(ref=True):
    """(Re)build libelmoracle.so and, when the reference is mounted, _ref/libelmref.so."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/src/physics"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


class _Lib:
    def __init__(self):
        # ELMO_LIBRARY: another build of the same sources (tests/test_oracle_sanitizers.py loads an ASan / UBSan build)
        p = os.environ.get("ELMO_LIBRARY") or os.path.join(HERE, "libelmoracle.so")
        if not os.path.exists(p):
            build(ref=False)
        self.lib = C.CDLL(p)
        L = self.lib
        L.elmo_create.restype = C.c_void_p
        L.elmo_create.argtypes = [C.c_int64]
        L.elmo_destroy.argtypes = [C.c_void_p]
        L.elmo_num_fields.restype = C.c_int
        L.elmo_field_name.restype = C.c_char_p
        L.elmo_field_name.argtypes = [C.c_int]
        L.elmo_field_ptr.restype = C.c_void_p
        L.elmo_field_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for n in ("snicar", "pft_psn", "pft_alb", "z0mr", "displar", "albsat", "albdry", "snowage"):
            f = getattr(L, f"elmo_{n}_ptr")
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        L.elmo_set_scalars.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_double, C.c_int, C.c_double, C.c_double]
        L.elmo_set_threads.argtypes = [C.c_int]
        L.elmo_get_max_threads.restype = C.c_int
        for n in ("frac_wet", "albedo_snicar", "surface_radiation", "canopy_temperature", "bareground_fluxes"):
            getattr(L, f"elmo_{n}").argtypes = [C.c_void_p]
        for n in ("canopy_hydrology", "canopy_fluxes", "timestep7"):
            getattr(L, f"elmo_{n}").argtypes = [C.c_void_p, C.c_double]
        L.elmo_canopy_fluxes_given.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 4
        L.elmo_bareground_fluxes_given.argtypes = [C.c_void_p, C.c_void_p]
        L.elmo_albedo_snicar_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.elmo_soil_temperature.argtypes = [C.c_void_p, C.c_double]
        L.elmo_snow_hydrology.argtypes = [C.c_void_p, C.c_double]
        L.elmo_snow_hydrology_stage.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.elmo_initialize_state.argtypes = [C.c_void_p]
        L.elmo_set_init_params.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.elmo_soil_temperature_ex.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 5
        L.elmo_soil_thermal.argtypes = [C.c_void_p] * 5
        L.elmo_surface_fluxes.argtypes = [C.c_void_p, C.c_double]
        L.elmo_init_timestep.argtypes = [C.c_void_p]
        L.elmo_get_forcing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.elmo_phenology.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.elmo_evaluate_conservation.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.elmo_pdma.argtypes = [C.c_int64] + [C.c_void_p] * 3
        L.elmo_phase_change.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        self.ref = _load(os.path.join(HERE, "_ref", "libelmref.so"))
        if self.ref is not None:
            R = self.ref
            for n in ("frac_wet", "surface_radiation", "canopy_temperature", "bareground_fluxes", "soil_moist_stress"):
                getattr(R, f"elmref_{n}").argtypes = [C.c_void_p]
            R.elmref_canopy_hydrology.argtypes = [C.c_void_p, C.c_double]
            R.elmref_snicar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            R.elmref_qsat.argtypes = [C.c_int64] + [C.c_void_p] * 6
            R.elmref_forc_derived.argtypes = [C.c_int64] + [C.c_void_p] * 6
            R.elmref_friction.argtypes = [C.c_int64] + [C.c_void_p] * 12
            if hasattr(R, "elmref_soil_thermal"):
                R.elmref_soil_thermal.argtypes = [C.c_void_p] * 5
                R.elmref_pdma.argtypes = [C.c_int64] + [C.c_void_p] * 3
                R.elmref_phase_change.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
            if hasattr(R, "elmref_surface_fluxes"):
                R.elmref_surface_fluxes.argtypes = [C.c_void_p, C.c_double]
            if hasattr(R, "elmref_init_timestep"):
                R.elmref_init_timestep.argtypes = [C.c_void_p]
                R.elmref_evaluate_conservation.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
            if hasattr(R, "elmref_initialize_state"):
                R.elmref_initialize_state.argtypes = [C.c_void_p]
            if hasattr(R, "elmref_get_forcing"):
                R.elmref_get_forcing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
                R.elmref_phenology.argtypes = [C.c_void_p, C.c_double, C.c_double]


        # the reference's snow-hydrology and soil-temperature functions: one library each (oracle/Makefile says why)
        self.ref_snow = _load(os.path.join(HERE, "_ref", "libelmref_snow.so"))
        if self.ref_snow is not None:
            self.ref_snow.elmref_snow_hydrology_stage.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p]
            self.ref_snow.elmref_snow_hydrology_stage.restype = C.c_int
            if hasattr(self.ref_snow, "elmref_aerosol_helpers"):
                self.ref_snow.elmref_aerosol_helpers.argtypes = [C.c_int64] + [C.c_void_p] * 6 + [C.c_double] + [C.c_void_p] * 2
                self.ref_snow.elmref_aerosol_helpers.restype = None
        self.ref_soil = _load(os.path.join(HERE, "_ref", "libelmref_soil.so"))
        if self.ref_soil is not None:
            self.ref_soil.elmref_soil_temperature.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 3
        # canopy_fluxes.h / photosynthesis.h / surface_albedo.h of the reference (ref_harness_canopy.cc says how they build here)
        self.ref_canopy = _load(os.path.join(HERE, "_ref", "libelmref_canopy.so"))
        if self.ref_canopy is not None:
            self.ref_canopy.elmref_canopy_fluxes.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 3
            self.ref_canopy.elmref_albedo_snicar.argtypes = [C.c_void_p] * 3
            self.ref_canopy.elmref_photosynthesis.argtypes = [C.c_int64] + [C.c_void_p] * 6


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def field_names():
    L = lib().lib
    return [L.elmo_field_name(i).decode() for i in range(L.elmo_num_fields())]


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleState:
    """Column state in the reference's [col][lev] layout; fields are numpy views into C memory."""

    def __init__(self, ncols):
        self._L = lib()
        self.ncols = int(ncols)
        self.ptr = self._L.lib.elmo_create(self.ncols)
        if not self.ptr:
            raise MemoryError("elmo_create failed")
        self.fields = {}
        self.nlev = {}
        for name in field_names() + ["err_flags"]:
            nlev, kind = C.c_int(), C.c_int()
            addr = self._L.lib.elmo_field_ptr(self.ptr, name.encode(), C.byref(nlev), C.byref(kind))
            dt = np.uint32 if name == "err_flags" else KIND_DTYPE[kind.value]
            shape = (self.ncols, nlev.value) if nlev.value > 1 else (self.ncols,)
            self.fields[name] = _view(addr, shape, dt) if self.ncols > 0 else np.zeros(shape, dt)
            self.nlev[name] = nlev.value
        L = self._L.lib
        self.pft_psn = _view(L.elmo_pft_psn_ptr(self.ptr), (25, 27), np.float64)
        self.pft_alb = _view(L.elmo_pft_alb_ptr(self.ptr), (25, 9), np.float64)
        self.z0mr = _view(L.elmo_z0mr_ptr(self.ptr), (25,), np.float64)
        self.displar = _view(L.elmo_displar_ptr(self.ptr), (25,), np.float64)
        self.albsat = _view(L.elmo_albsat_ptr(self.ptr), (20, 2), np.float64)
        self.albdry = _view(L.elmo_albdry_ptr(self.ptr), (20, 2), np.float64)
        # SnwRdsTable (snicar_data.h:75-84): snowage_tau / kappa / drdt0, each [11, 31, 8]
        self.snowage = _view(L.elmo_snowage_ptr(self.ptr), (3, 11, 31, 8), np.float64)
        self.snicar = {}
        off = L.elmo_snicar_ptr(self.ptr)
        for name, n in SNICAR_TABLES:
            self.snicar[name] = _view(off, (n,), np.float64)
            off += n * 8
        self.scalars = dict(ltype=1, ctype=0, vtype=2, urbpoi=0, lakpoi=0, dewmx=0.1, oldfflag=1, dayl=0.0, max_dayl=0.0)
        self.init_params = None

    def __del__(self):
        try:
            if self.ptr:
                self._L.lib.elmo_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def __getitem__(self, name):
        return self.fields[name]

    def set_scalars(self, **kw):
        self.scalars.update(kw)
        s = self.scalars
        self._L.lib.elmo_set_scalars(
            self.ptr, int(s["ltype"]), int(s["ctype"]), int(s["vtype"]), int(s["urbpoi"]), int(s["lakpoi"]),
            float(s["dewmx"]), int(s["oldfflag"]), float(s["dayl"]), float(s["max_dayl"]),
        )

    def load_params(self, pft=None, optics=None):
        """Fill the shared tables from the committed fixtures (tests/golden/pft_params.npz, SnowOptics.npz)."""
        pft = pft if pft is not None else np.load(os.path.join(GOLDEN, "pft_params.npz"))
        optics = optics if optics is not None else np.load(os.path.join(GOLDEN, "SnowOptics.npz"))
        set_pft_tables(self.pft_psn, self.pft_alb, self.z0mr, self.displar, pft)
        for name, n in SNICAR_TABLES:
            self.snicar[name][:] = np.asarray(optics[name], dtype=np.float64).reshape(-1)[:n]

    def copy_from(self, other):
        for k, v in other.fields.items():
            self.fields[k][...] = v
        self.pft_psn[...] = other.pft_psn
        self.pft_alb[...] = other.pft_alb
        self.z0mr[...] = other.z0mr
        self.displar[...] = other.displar
        self.albsat[...] = other.albsat
        self.albdry[...] = other.albdry
        for k in self.snicar:
            self.snicar[k][...] = other.snicar[k]
        self.snowage[...] = other.snowage
        if getattr(other, "init_params", None) is not None:
            self.set_init_params(*other.init_params)
        self.set_scalars(**other.scalars)

    def clone(self):
        o = OracleState(self.ncols)
        o.copy_from(self)
        return o

    # -- the L3 wrappers ------------------------------------------------------------------------
    def frac_wet(self):
        self._L.lib.elmo_frac_wet(self.ptr)

    def albedo_snicar(self):
        self._L.lib.elmo_albedo_snicar(self.ptr)

    def canopy_hydrology(self, dt):
        self._L.lib.elmo_canopy_hydrology(self.ptr, float(dt))

    def surface_radiation(self):
        self._L.lib.elmo_surface_radiation(self.ptr)

    def canopy_temperature(self):
        self._L.lib.elmo_canopy_temperature(self.ptr)

    def bareground_fluxes(self):
        self._L.lib.elmo_bareground_fluxes(self.ptr)

    def canopy_fluxes(self, dt):
        self._L.lib.elmo_canopy_fluxes(self.ptr, float(dt))

    def bareground_fluxes_given(self, rho):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        self._L.lib.elmo_bareground_fluxes_given(self.ptr, rho.ctypes.data)

    def albedo_snicar_ex(self):
        """-> (fabd_sun, fabd_sha): the wrapper-local arrays the reference never stores in the state."""
        sun = np.zeros((self.ncols, 2))
        sha = np.zeros((self.ncols, 2))
        self._L.lib.elmo_albedo_snicar_ex(self.ptr, sun.ctypes.data, sha.ctypes.data)
        return sun, sha

    def canopy_fluxes_given(self, dt, rho=None, po2=None, pco2=None, want_niter=False):
        """L2-level entry: forcing-derived scalars handed in (as test_CanFlux.cc does) instead of derived."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (rho, po2, pco2)]
        niter = np.zeros(self.ncols, dtype=np.int32) if want_niter else None
        self._L.lib.elmo_canopy_fluxes_given(
            self.ptr, float(dt), *[None if a is None else a.ctypes.data for a in arrs],
            None if niter is None else niter.ctypes.data,
        )
        return niter

    def timestep7(self, dt):
        self._L.lib.elmo_timestep7(self.ptr, float(dt))

    # -- next row: soil / snow temperature (soil_temperature_kokkos.cc) ----------------------------
    def soil_temperature(self, dt):
        self._L.lib.elmo_soil_temperature(self.ptr, float(dt))

    def soil_temperature_ex(self, dt):
        """-> dict(lhs [n,21,5], rhs [n,21] before the solve, sol [n,21], cv [n,20], hs [n,4] = hs_soil, hs_h2osfc,
        hs_top_snow, dhsdT): the wrapper-local system of each column."""
        n = self.ncols
        out = dict(lhs=np.zeros((n, 21, 5)), rhs=np.zeros((n, 21)), sol=np.zeros((n, 21)), cv=np.zeros((n, 20)),
                   hs=np.zeros((n, 4)))
        self._L.lib.elmo_soil_temperature_ex(self.ptr, float(dt), *[out[k].ctypes.data for k in ("lhs", "rhs", "sol", "cv", "hs")])
        return out

    def snow_hydrology(self, dt):
        """kokkos_snow_hydrology (snow_hydrology_kokkos.cc:23-188); oracle/elmo_physics_g.c says what pins it."""
        self._L.lib.elmo_snow_hydrology(self.ptr, float(dt))

    SNOW_STAGES = ("snow_water", "aerosol_deposition", "aerosol_phase_change", "transpiration", "snow_compaction",
                   "combine_layers", "divide_layers", "prune_snow_layers", "aerosol_mass_and_concen", "snow_aging")
    SNOW_STAGES_REF = (0, 2, 3, 4, 5, 6, 7, 9)  # the stages the reference's own functions can run here (ref_harness_snow.cc)

    def snow_hydrology_stage(self, dt, stage, ref=False, skip=None):
        """One stage of the wrapper over all columns; ref=True: the reference's own function (columns with skip != 0 untouched).
        Returns the number of columns in which the reference threw (0 for the restatement)."""
        if not ref:
            self._L.lib.elmo_snow_hydrology_stage(self.ptr, float(dt), int(stage))
            return 0
        sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
        n = self._L.ref_snow.elmref_snow_hydrology_stage(self.ptr, float(dt), int(stage), None if sk is None else sk.ctypes.data)
        assert n >= 0, "stage has no reference run"
        return n

    def soil_temperature_ref(self, dt):
        """kokkos_soil_temperature by the reference's own per-column functions (ref_harness_soil.cc); returns lhs / rhs / hs as
        soil_temperature_ex does."""
        n = self.ncols
        out = dict(lhs=np.zeros((n, 21, 5)), rhs=np.zeros((n, 21)), hs=np.zeros((n, 4)))
        self._L.ref_soil.elmref_soil_temperature(self.ptr, float(dt), *[out[k].ctypes.data for k in ("lhs", "rhs", "hs")])
        return out

    def canopy_fluxes_ref(self, dt, rho=None, po2=None, pco2=None):
        """kokkos_canopy_fluxes by the reference's own initialize_flux / stability_iteration (photosynthesis inside) /
        compute_flux (ref_harness_canopy.cc); a column in which the reference threw gets bit 31 of err_flags."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (rho, po2, pco2)]
        self._L.ref_canopy.elmref_canopy_fluxes(self.ptr, float(dt), *[None if a is None else a.ctypes.data for a in arrs])

    def albedo_snicar_ref(self):
        """kokkos_albedo_snicar by the reference's own surface_albedo / snow_snicar functions -> (fabd_sun, fabd_sha)."""
        sun = np.zeros((self.ncols, 2))
        sha = np.zeros((self.ncols, 2))
        self._L.ref_canopy.elmref_albedo_snicar(self.ptr, sun.ctypes.data, sha.ctypes.data)
        return sun, sha

    def set_init_params(self, organic_max, roota_par, rootb_par):
        """organic_max of the parameter file (initialize_elm_kokkos.cc:312) and PFTData::roota_par / rootb_par [25]."""
        a = np.ascontiguousarray(roota_par, dtype=np.float64)
        b = np.ascontiguousarray(rootb_par, dtype=np.float64)
        assert a.shape == (25,) and b.shape == (25,)
        self.init_params = (float(organic_max), a.copy(), b.copy())
        self._L.lib.elmo_set_init_params(self.ptr, float(organic_max), a.ctypes.data, b.ctypes.data)

    def initialize_state(self, lib=None):
        """The per-column init functions of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428)."""
        (self._L.lib.elmo_initialize_state if lib is None else lib.elmref_initialize_state)(self.ptr)

    def init_timestep(self, lib=None):
        (self._L.lib.elmo_init_timestep if lib is None else lib.elmref_init_timestep)(self.ptr)

    def get_forcing(self, wt1, wt2, qbot_is_rh=False, lib=None):
        """get_forcing (atm_forcing_kokkos.cc:47-75); wt1, wt2: [8] weights of TBOT, PBOT, QBOT|RH, FLDS, FSDS, PREC, WIND, ZBOT."""
        w1 = np.ascontiguousarray(wt1, dtype=np.float64)
        w2 = np.ascontiguousarray(wt2, dtype=np.float64)
        assert w1.shape == (8,) and w2.shape == (8,)
        (self._L.lib.elmo_get_forcing if lib is None else lib.elmref_get_forcing)(self.ptr, w1.ctypes.data, w2.ctypes.data, int(bool(qbot_is_rh)))

    def phenology(self, wt1, wt2, lib=None):
        (self._L.lib.elmo_phenology if lib is None else lib.elmref_phenology)(self.ptr, float(wt1), float(wt2))

    def surface_fluxes(self, dt, lib=None):
        (self._L.lib.elmo_surface_fluxes if lib is None else lib.elmref_surface_fluxes)(self.ptr, float(dt))

    DIAG_NAMES = ("dtend_column_h2o", "errh2o", "errh2osno", "dwb", "errsol", "errlon", "errseb", "netrad")

    def evaluate_conservation(self, dt, lib=None):
        """-> [ncols, 8] (DIAG_NAMES): the wrapper-local diagnostics of kokkos_evaluate_conservation."""
        d = np.zeros((self.ncols, 8))
        (self._L.lib.elmo_evaluate_conservation if lib is None else lib.elmref_evaluate_conservation)(self.ptr, float(dt), d.ctypes.data)
        return d

    def soil_thermal(self, lib=None):
        """-> (thk, tk, cv [n,20], scal [n,3] = tk_h2osfc, c_h2osfc, dz_h2osfc); lib: the Reference's library instead."""
        n = self.ncols
        thk, tk, cv, scal = np.zeros((n, 20)), np.zeros((n, 20)), np.zeros((n, 20)), np.zeros((n, 3))
        fn = self._L.lib.elmo_soil_thermal if lib is None else lib.elmref_soil_thermal
        fn(self.ptr, thk.ctypes.data, tk.ctypes.data, cv.ctypes.data, scal.ctypes.data)
        return thk, tk, cv, scal

    def phase_change(self, dt, dhsdT, c_h2osfc, lib=None):
        a = np.ascontiguousarray(dhsdT, dtype=np.float64)
        b = np.ascontiguousarray(c_h2osfc, dtype=np.float64)
        fn = self._L.lib.elmo_phase_change if lib is None else lib.elmref_phase_change
        fn(self.ptr, float(dt), a.ctypes.data, b.ctypes.data)


def pdma(snl, lhs, rhs, lib=None):
    """Solve the pentadiagonal systems (lhs [n,21,5], rhs [n,21]) -> solution [n,21]; lib: the Reference's library."""
    snl = np.ascontiguousarray(snl, dtype=np.int32)
    lhs = np.ascontiguousarray(lhs, dtype=np.float64)
    sol = np.array(rhs, dtype=np.float64, order="C", copy=True)
    fn = globals()["lib"]().lib.elmo_pdma if lib is None else lib.elmref_pdma
    fn(snl.shape[0], snl.ctypes.data, lhs.ctypes.data, sol.ctypes.data)
    return sol


def set_pft_tables(psn, alb, z0mr, displar, pft):
    """PFTData::get_pft_psn / get_pft_alb wiring (src/data/pft_data_impl.hh:64-116) for all 25 PFTs."""
    for j, name in enumerate(PSN_FIELDS):
        v = np.asarray(pft[name], dtype=np.float64).reshape(-1)
        psn[:, j] = v[0] if name == "tc_stress" else v[:25]
    for j, name in enumerate(
        ["rholvis", "rholnir", "rhosvis", "rhosnir", "taulvis", "taulnir", "tausvis", "tausnir", "xl"]
    ):
        alb[:, j] = np.asarray(pft[name], dtype=np.float64).reshape(-1)[:25]
    z0mr[:] = np.asarray(pft["z0mr"]).reshape(-1)[:25]
    displar[:] = np.asarray(pft["displar"]).reshape(-1)[:25]


def psn_counters(reset=False):
    """Branch counters of the restatement's photosynthesis root find: dict(hybrid, brent, itmax, c4)."""
    out = (C.c_ulonglong * 4)()
    lib().lib.elmo_psn_counters(out, 1 if reset else 0)
    return dict(zip(("hybrid", "brent", "itmax", "c4"), [int(v) for v in out]))


def aerosol_helpers_ref(snow_idx, snotop, do_capsnow, h2osoi_ice, h2osoi_liq, qflx_snwcp_ice, dtime):
    """ELM::aero_impl::get_snow_mass / get_snowcap_scl_fct (aerosol_physics_impl.hh:10-31) - the reference's own, compiled in
    oracle/_ref/libelmref_snow.so - elementwise over equally shaped arrays -> (snowmass, snowcap_scl_fct)."""
    a = [np.ascontiguousarray(x, dtype=t) for x, t in ((snow_idx, np.int32), (snotop, np.int32), (do_capsnow, np.int32),
                                                      (h2osoi_ice, np.float64), (h2osoi_liq, np.float64), (qflx_snwcp_ice, np.float64))]
    assert all(x.shape == a[0].shape for x in a)
    mass, scl = np.zeros(a[0].shape), np.zeros(a[0].shape)
    lib().ref_snow.elmref_aerosol_helpers(a[0].size, *[x.ctypes.data for x in a], float(dtime), mass.ctypes.data, scl.ctypes.data)
    return mass, scl


def have_ref_canopy():
    return lib().ref_canopy is not None


def have_ref():
    return lib().ref is not None


class Reference:
    """The compiled reference headers (oracle/_ref/libelmref.so) driven on an OracleState."""

    def __init__(self):
        self.R = lib().ref
        if self.R is None:
            raise RuntimeError("oracle/_ref/libelmref.so not built (reference not mounted here)")

    def frac_wet(self, S):
        self.R.elmref_frac_wet(S.ptr)

    def canopy_hydrology(self, S, dt):
        self.R.elmref_canopy_hydrology(S.ptr, float(dt))

    def surface_radiation(self, S):
        self.R.elmref_surface_radiation(S.ptr)

    def canopy_temperature(self, S):
        self.R.elmref_canopy_temperature(S.ptr)

    def bareground_fluxes(self, S):
        self.R.elmref_bareground_fluxes(S.ptr)

    def soil_moist_stress(self, S):
        self.R.elmref_soil_moist_stress(S.ptr)

    def snicar(self, S):
        d = np.zeros((S.ncols, 6, 2))
        i = np.zeros((S.ncols, 6, 2))
        self.R.elmref_snicar(S.ptr, d.ctypes.data, i.ctypes.data)
        return d, i

    def qsat(self, T, p):
        T = np.ascontiguousarray(T, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = [np.zeros_like(T) for _ in range(4)]
        self.R.elmref_qsat(T.size, T.ctypes.data, p.ctypes.data, *[o.ctypes.data for o in out])
        return out

    def forc_derived(self, pbot, qbot, tbot):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (pbot, qbot, tbot)]
        out = [np.zeros_like(a[0]) for _ in range(3)]
        self.R.elmref_forc_derived(a[0].size, *[x.ctypes.data for x in a], *[o.ctypes.data for o in out])
        return out

    def friction(self, **kw):
        names = "ur thv dthv zldis z0m z0h z0q hgt_u hgt_t hgt_q displa".split()
        a = [np.ascontiguousarray(kw[n], dtype=np.float64) for n in names]
        out = np.zeros((a[0].size, 7))
        self.R.elmref_friction(a[0].size, *[x.ctypes.data for x in a], out.ctypes.data)
        return out
