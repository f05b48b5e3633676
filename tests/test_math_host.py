"""elmkernels_amd/csrc/elmk_math.h (the device's exp / log / log10 / pow / atan / expm1 / tanh / cos / erf / acos) compiled for the host with gcc and compared
with the live libm - the one the oracle and the reference call - bit for bit.  The header's purpose and provenance are in
its own comment; tools/gen_libm_tables.py writes the table file it includes."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "tools", "math_host_check.c")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("mathchk") / "math_host_check")
    # -ffp-contract=off: only the explicit fma() calls of the header may be fused; -mfma: they are one instruction
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", SRC, "-o", exe, "-lm"])
    return exe


@pytest.mark.parametrize("seed", [1, 20261004])
def test_host_build_of_device_math_matches_libm_bit_for_bit(checker, seed):
    """6 argument classes x 2 M arguments (x 1-3 variants) per function (physics ranges, whole exponent range, random bit patterns,
    subnormals, over/underflow edges, the exponents the physics uses): zero mismatches."""
    r = subprocess.run([checker, "2000000", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if "mismatches=" in ln]
    assert len(lines) == 10 and all(ln.endswith("mismatches=0") for ln in lines), r.stdout


def test_table_header_is_what_the_live_libm_holds(tmp_path):
    """The committed elmk_math_tables.h equals what tools/gen_libm_tables.py extracts from this machine's libm.so.6."""
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no glibc libm at the usual place")
    gen = os.path.join(ROOT, "tools", "gen_libm_tables.py")
    committed = open(os.path.join(ROOT, "elmkernels_amd", "csrc", "elmk_math_tables.h")).read()
    env = dict(os.environ, ELMK_TABLES_OUT=str(tmp_path / "t.h"))
    subprocess.check_call([sys.executable, gen, libm], env=env)
    assert open(tmp_path / "t.h").read() == committed
