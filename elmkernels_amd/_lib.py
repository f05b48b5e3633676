"""ctypes binding of libelmk.so (the C ABI of include/elmk.h).

The library is built in-tree (elmkernels_amd/csrc/Makefile, or __graft_entry__.build()).  There is no
Python/CPU fallback: if the shared object is missing, or no MI355X-class HIP device is usable, the physics
entry points raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ELMK_LIBRARY selects another build of the same ABI (development: kernel variants under test)
LIB_PATH = os.environ.get("ELMK_LIBRARY") or os.path.join(HERE, "libelmk.so")

# every symbol include/elmk.h declares: (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "elmk_create": (C.c_int, [C.c_int64, C.c_int, C.POINTER(_P)]),
    "elmk_destroy": (C.c_int, [_P]),
    "elmk_last_error": (C.c_char_p, [_P]),
    "elmk_set_stream": (C.c_int, [_P, _P]),
    "elmk_sync": (C.c_int, [_P]),
    "elmk_ncols": (C.c_int64, [_P]),
    "elmk_level_stride": (C.c_int64, [_P]),
    "elmk_device_bytes": (C.c_int64, [_P]),
    "elmk_num_fields": (C.c_int, []),
    "elmk_field_name": (C.c_char_p, [C.c_int]),
    "elmk_field_id": (C.c_int, [C.c_char_p]),
    "elmk_field_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "elmk_upload": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int]),
    "elmk_download": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int]),
    "elmk_fill": (C.c_int, [_P, C.c_int, C.c_double]),
    "elmk_device_ptr": (_P, [_P, C.c_int]),
    "elmk_tile_columns": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_int, _P]),
    "elmk_snapshot_fields": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    "elmk_restore_fields": (C.c_int, [_P]),
    "elmk_set_land": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "elmk_set_scalars": (C.c_int, [_P, C.c_double, C.c_int, C.c_double, C.c_double]),
    "elmk_set_pft": (C.c_int, [_P, _P, _P, _P, _P]),
    "elmk_set_soilcolor": (C.c_int, [_P, _P, _P]),
    "elmk_set_snicar": (C.c_int, [_P, _P]),
    "elmk_frac_wet": (C.c_int, [_P]),
    "elmk_albedo_snicar": (C.c_int, [_P]),
    "elmk_canopy_hydrology": (C.c_int, [_P, C.c_double]),
    "elmk_surface_radiation": (C.c_int, [_P]),
    "elmk_canopy_temperature": (C.c_int, [_P]),
    "elmk_bareground_fluxes": (C.c_int, [_P]),
    "elmk_canopy_fluxes": (C.c_int, [_P, C.c_double]),
    "elmk_canopy_fluxes_given": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "elmk_bareground_fluxes_given": (C.c_int, [_P, _P]),
    "elmk_timestep7": (C.c_int, [_P, C.c_double]),
    "elmk_timestep7_fused": (C.c_int, [_P, C.c_double]),
    "elmk_advance_physics": (C.c_int, [_P, C.c_double]),
    "elmk_initialize_state": (C.c_int, [_P]),
    "elmk_set_init_params": (C.c_int, [_P, C.c_double, _P, _P]),
    "elmk_profile_timestep7_fused": (C.c_int, [_P, C.c_double, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "elmk_soil_temperature": (C.c_int, [_P, C.c_double]),
    "elmk_snow_hydrology": (C.c_int, [_P, C.c_double]),
    "elmk_set_snow_age_tables": (C.c_int, [_P, _P, _P, _P]),
    "elmk_surface_fluxes": (C.c_int, [_P, C.c_double]),
    "elmk_init_timestep": (C.c_int, [_P]),
    "elmk_set_graph": (C.c_int, [_P, C.c_int]),
    "elmk_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "elmk_get_forcing": (C.c_int, [_P, _P, _P, C.c_int]),
    "elmk_phenology": (C.c_int, [_P, C.c_double, C.c_double]),
    "elmk_evaluate_conservation": (C.c_int, [_P, C.c_double, _P, _P]),
    "elmk_error_summary": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]),
    "elmk_clear_errors": (C.c_int, [_P]),
    "elmk_profile_timestep7": (C.c_int, [_P, C.c_double, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "elmk_profile_wrapper": (C.c_int, [_P, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_float)]),
    "elmk_read_scratch": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64]),
    "elmk_state_real_bytes": (C.c_int, []),
    "elmk_profile_steps": (C.c_int, [_P, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_float)]),
    "elmk_copy_bandwidth": (C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    "elmk_copy_bandwidth_shape": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "elmk_math_eval": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64]),
}

# ELM::SnicarData member order as laid out in elmk_snicar_tables (include/elmk.h)
SNICAR_NAMES = (
    [f"{p}_{s}" for s in ("oc1", "oc2", "dst1", "dst2", "dst3", "dst4") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + [f"{p}_snw_{s}" for s in ("drc", "dfs") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + [f"{p}_{s}" for s in ("bc1", "bc2") for p in ("ss_alb", "asm_prm", "ext_cff_mss")]
    + ["bcenh"]
)
SNICAR_SIZES = dict(
    [(n, 5) for n in SNICAR_NAMES[:18]] + [(n, 5 * 1471) for n in SNICAR_NAMES[18:24]]
    + [(n, 50) for n in SNICAR_NAMES[24:30]] + [("bcenh", 400)]
)


class SnicarTables(C.Structure):
    _fields_ = [(n, _P) for n in SNICAR_NAMES]


class Perturb(C.Structure):
    _fields_ = [("field", C.c_int32), ("mode", C.c_int32), ("amp", C.c_double)]


class ElmkError(RuntimeError):
    pass


_libs = {}
# the report-only build of BASELINE config 5 (every fp64 state field stored as fp32, same ABI): never the default
F32_LIB_PATH = os.path.join(HERE, "libelmk_f32.so")


def load(path=None):
    """Load libelmk.so (or another build of the same ABI) and declare every entry point; fails loudly when the HIP
    extension is not built."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ElmkError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C elmkernels_amd/csrc). "
            "elmkernels_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale w.r.t. include/elmk.h
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib
