"""Reader of the reference's on-disk single-column text format - the only data format it bundles (test/data/*_IN.txt,
*_OUT.txt) - mirroring ELM::IO::ELMtestinput (src/utils/read_test_input.hh:27-101, read_test_input.cc:14-43):

    NSTEP <n>
    <label> v0 v1 ...
    ...
    !!! <n>

`get_state(n)` cuts the block between "NSTEP n\\n" and "!!! n\\n"; `parse_state(label, size)` returns the values of
the first line of that block whose first token equals the label and raises if their number differs from `size` or the
label is missing (the reference throws std::runtime_error with the same messages, :53-57, :66-67);
`compare_output(label, values, rel_tol)` is compareOutput with the reference's IsAlmostEqual (:17-24) - but it RETURNS the
verdict instead of only printing it (the reference's tests always exit 0).  `upload_state` feeds every labelled line of a
block that names a state field into an ELMState: one block = one column, as in the reference's tests, so a file of N
blocks drives N columns (or N steps of one column).  Values may be `nan` and `1e+36` sentinels.
"""
import numpy as np


def is_almost_equal(a, b, rel_tol=1e-15, abs_tol=1e-20):
    """read_test_input.hh:17-24 (elementwise; nan equals nothing, as in the reference)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        diff = np.abs(a - b)
        return (diff <= np.maximum(np.abs(a), np.abs(b)) * rel_tol) | (diff <= abs_tol)


class ELMtestinput:
    def __init__(self, filename):
        self.filename = filename
        try:
            with open(filename) as fh:
                self.filestring = fh.read()
        except OSError as e:
            raise RuntimeError(f"INPUT ERROR: Can't open input file {filename}") from e
        self.state = ""
        self.nstep = 0

    def steps(self):
        """Every NSTEP id in file order (the reference's tests hard-wire their ranges)."""
        return [int(ln.split()[1]) for ln in self.filestring.splitlines() if ln.startswith("NSTEP ")]

    def get_state(self, nstep):
        self.nstep = int(nstep)
        start = self.filestring.find(f"NSTEP {self.nstep}\n")
        end = self.filestring.find(f"!!! {self.nstep}\n")
        if start < 0 or end < 0:
            raise RuntimeError(f"INPUT ERROR: no block NSTEP {self.nstep} in {self.filename}")
        self.state = self.filestring[start:end]
        return self.state

    def labels(self):
        return [ln.split()[0] for ln in self.state.splitlines()[1:] if ln.split()]

    def parse_state(self, label, size=None, dtype=np.float64):
        for line in self.state.splitlines():
            tok = line.split()
            if tok and tok[0] == label:
                if size is not None and len(tok) - 1 != int(size):
                    raise RuntimeError(f"INPUT ERROR: Array length ({int(size)}) != input data length ({len(tok) - 1}) "
                                       f"for variable {label}")
                vals = np.array([float(v) for v in tok[1:]], dtype=np.float64)
                return vals if dtype == np.float64 else np.nan_to_num(vals, nan=0.0).astype(dtype)
        raise RuntimeError(f"INPUT ERROR: Can't find variable {label} in NSTEP {self.nstep}")

    def compare_output(self, label, values, rel_tol=1e-15):
        """-> (passes, [(index, ours, file), ...]) like compareOutput's printout (:70-89)."""
        values = np.asarray(values, dtype=np.float64).ravel()
        ref = self.parse_state(label, values.size)
        ok = is_almost_equal(values, ref, rel_tol)
        return bool(ok.all()), [(int(i), float(values[i]), float(ref[i])) for i in np.nonzero(~ok)[0]]


def upload_state(S, inp, steps, rename=None):
    """Blocks `steps` of `inp` (an ELMtestinput) -> columns 0..len(steps)-1 of the ELMState S, for every label that is a
    state field (after `rename`, e.g. {"forc_t": "forc_tbot"}: the reference's wrappers pass S.forc_tbot as forc_t,
    SURVEY 8a quirk 11).  Returns the labels that were not state fields."""
    rename = rename or {}
    steps = list(steps)
    assert len(steps) == S.ncols
    per_field, skipped = {}, set()
    for col, n in enumerate(steps):
        inp.get_state(n)
        for label in inp.labels():
            name = rename.get(label, label)
            if name not in S.fields:
                skipped.add(label)
                continue
            fid, nlev, dt = S.fields[name]
            vals = inp.parse_state(label, nlev, dtype=dt)
            per_field.setdefault(name, np.zeros((len(steps), nlev), dtype=dt))[col] = vals
    for name, arr in per_field.items():
        S.upload(name, arr if arr.shape[1] > 1 else arr[:, 0])
    return sorted(skipped)
