"""Golden-fixture plumbing shared by the oracle tests (CPU) and the HIP parity tests (GPU).

The reference's tests (test/test_*.cc) parse one NSTEP block at a time into 1-column arrays, call the
L2 physics and compare against the _OUT block with IsAlmostEqual(rel 1e-15, abs 1e-20)
(src/utils/read_test_input.hh:17-24,70-89).  Here every fixture step becomes ONE COLUMN of a state,
so a whole fixture runs as a single wrapper call over nsteps columns.
"""
import contextlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fixture label -> state field (labels that are already field names map to themselves)
RENAME = {
    "forc_t": "forc_tbot",
    "forc_q": "forc_qbot",
    "forc_th": "forc_thbot",
    "z": "zsoi",
    "zi": "zisoi",
    "mss_cnc_bcphi": "cnc_bcphi",
    "mss_cnc_bcpho": "cnc_bcpho",
    "mss_cnc_dst1": "cnc_dst1",
    "mss_cnc_dst2": "cnc_dst2",
    "mss_cnc_dst3": "cnc_dst3",
    "mss_cnc_dst4": "cnc_dst4",
    "albsnd_hst": "albsnd",
    "albsni_hst": "albsni",
}

# the hard-wired LandType / dtime of every reference test (e.g. test/test_CanHydro.cc:96-103)
TEST_LAND = dict(ltype=1, ctype=1, vtype=12, urbpoi=0, lakpoi=0)
TEST_DTIME = 1800.0

# step ranges the reference tests loop over (test_CanHydro.cc:150, test_CanFlux.cc:280, test_SurfAlb.cc:392)
STEP_RANGE = {
    "CanopyHydrology": (1, 49),
    "CanopySunShadeFractions": (1, 49),
    "SurfaceRadiation": (1, 49),
    "CanopyTemperature": (1, 49),
    "BareGroundFluxes": (1, 49),
    "CanopyFluxes": (0, 97),
    "SurfaceAlbedo": (2, 49),
}


# Which of the reference's two fixture dumps load() serves: "data" = test/data, the one its tests read; "newdata" =
# test/new_data, a second ELM dump of the same site under snow-free summer forcing that no reference test reads (97 steps
# per module; its BareGroundFluxes_IN.txt is malformed - duplicate keys - and is not used).  tests/golden/make_golden.py
# converts both.
_DATASET = "data"


@contextlib.contextmanager
def dataset(name):
    global _DATASET
    old, _DATASET = _DATASET, name
    try:
        yield
    finally:
        _DATASET = old


def load(module):
    return np.load(os.path.join(GOLDEN, "" if _DATASET == "data" else _DATASET, module + ".npz"))


def select_steps(d, module, all_steps=False):
    steps = d["steps"]
    if all_steps or _DATASET != "data":
        return np.arange(len(steps))
    lo, hi = STEP_RANGE[module]
    return np.nonzero((steps >= lo) & (steps < hi))[0]


def split(d, prefix, rows, field_nlev):
    """-> (state fields {name: [n, nlev]}, other labels {label: [n, k]}) for 'in/' or 'out/'."""
    fields, other = {}, {}
    for key in d.files:
        if not key.startswith(prefix):
            continue
        label = key[len(prefix):]
        arr = d[key][rows]
        name = RENAME.get(label, label)
        if name in field_nlev and arr.shape[1] == field_nlev[name]:
            fields[name] = arr
        else:
            other[label] = arr
    return fields, other


def almost_equal(a, b, rel=1e-15, abs_tol=1e-20):
    """Element-wise ELM::IO::IsAlmostEqual; NaN never equals (as in C++), so callers mask NaN expectations."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        diff = np.abs(a - b)
        return (a == b) | (diff <= np.maximum(np.abs(a), np.abs(b)) * rel) | (diff <= abs_tol)


def rel_err(a, b, floor=1e-18):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.abs(a - b)
        den = np.maximum(np.abs(a), np.abs(b))
        r = np.where(d <= floor, 0.0, d / np.where(den > 0, den, 1.0))
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (a == b)
    return np.where(both_nan | same_inf | (a == b), 0.0, r)


def fill_state(S, fields):
    """Copy {name: [n, nlev]} into a state exposing .fields numpy views (OracleState) - ints are cast."""
    for name, arr in fields.items():
        dst = S.fields[name]
        src = arr if dst.ndim == 2 else arr[:, 0]
        if dst.dtype.kind in "iu":
            dst[...] = np.nan_to_num(src, nan=0.0).astype(dst.dtype)
        else:
            dst[...] = src
