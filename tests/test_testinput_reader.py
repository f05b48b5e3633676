"""elmkernels_amd/testinput.py - the reader of the reference's NSTEP-block text format (ELM::IO::ELMtestinput,
src/utils/read_test_input.hh / .cc) - on a file written here in that format, and, where the reference tree is mounted
(build container only), on one of its own fixture files against the committed golden conversion."""
import os

import numpy as np
import pytest

from elmkernels_amd import testinput as TI
from tests import fixtures as F

TEXT = """NSTEP 1
snl 0
t_soisno 270.5 271 nan 1e+36 273.15
forc_rain 1.5e-05
forc_rain 9.0
!!! 1
NSTEP 12
snl -2
t_soisno 1 2 3 4 5
forc_rain 0
!!! 12
"""


@pytest.fixture()
def path(tmp_path):
    p = tmp_path / "Mod_IN.txt"
    p.write_text(TEXT)
    return str(p)


def test_blocks_labels_and_values(path):
    inp = TI.ELMtestinput(path)
    assert inp.steps() == [1, 12]
    inp.get_state(12)
    assert inp.labels() == ["snl", "t_soisno", "forc_rain"]
    assert np.array_equal(inp.parse_state("t_soisno", 5), [1, 2, 3, 4, 5])
    assert inp.parse_state("snl", 1, dtype=np.int32)[0] == -2
    inp.get_state(1)  # "NSTEP 1" + newline must not match inside "NSTEP 12\\n"
    v = inp.parse_state("t_soisno", 5)
    assert v[0] == 270.5 and np.isnan(v[2]) and v[3] == 1e36
    assert inp.parse_state("forc_rain", 1)[0] == 1.5e-05  # the first matching line wins (:47-62)


def test_errors_are_the_references(path):
    inp = TI.ELMtestinput(path)
    inp.get_state(1)
    with pytest.raises(RuntimeError, match=r"Array length \(4\) != input data length \(5\) for variable t_soisno"):
        inp.parse_state("t_soisno", 4)
    with pytest.raises(RuntimeError, match="Can't find variable h2ocan in NSTEP 1"):
        inp.parse_state("h2ocan", 1)
    with pytest.raises(RuntimeError, match="Can't open input file"):
        TI.ELMtestinput(path + ".missing")


def test_compare_output_is_almost_equal(path):
    inp = TI.ELMtestinput(path)
    inp.get_state(12)
    ok, bad = inp.compare_output("t_soisno", [1, 2, 3 * (1 + 5e-16), 4, 5])
    assert ok and not bad
    ok, bad = inp.compare_output("t_soisno", [1, 2, 3 * (1 + 5e-15), 4, 5])
    assert not ok and bad[0][0] == 2
    assert TI.is_almost_equal(0.0, 5e-21).all() and not TI.is_almost_equal(0.0, 5e-20).all()  # the 1e-20 floor
    assert not TI.is_almost_equal(np.nan, np.nan).any()


@pytest.mark.skipif(not os.path.exists("/root/reference/test/data/CanopyHydrology_IN.txt"), reason="reference tree not mounted")
def test_reads_a_reference_fixture_like_the_golden_conversion():
    inp = TI.ELMtestinput("/root/reference/test/data/CanopyHydrology_IN.txt")
    d = F.load("CanopyHydrology")
    steps = [int(s) for s in d["steps"]]
    assert inp.steps() == steps
    for i in (0, len(steps) // 2, len(steps) - 1):
        inp.get_state(steps[i])
        for label in inp.labels():
            want = d["in/" + label][i]
            got = inp.parse_state(label, want.size)
            assert np.array_equal(got, want, equal_nan=True), (steps[i], label)
