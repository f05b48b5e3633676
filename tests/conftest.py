import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _detect_host_libm()


def _detect_host_libm():
    """Is the host libm the one elmk_math.h restates (glibc 2.35, x86-64 FMA variants)?  The header is compiled for the host
    and compared with the live libm on 20 000 arguments per class and function (tests/tools/math_host_check.c; the full
    2 x 10^7-argument comparison is tests/test_math_host.py).  The answer selects the parity bar: tests/_parity_mode.py."""
    import shutil
    import subprocess
    import tempfile

    from tests import _parity_mode as M

    src = os.path.join(ROOT, "tests", "tools", "math_host_check.c")
    if shutil.which("gcc") is None:
        M.BITWISE_VALID, M.REASON = False, "no gcc to build the host libm check: parity bar 1e-12"
        return
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "chk")
        try:
            subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", src, "-o", exe, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            r = subprocess.run([exe, "20000", "7"], capture_output=True, text=True, timeout=120)
            lines = [ln for ln in r.stdout.splitlines() if "mismatches=" in ln]
            ok = r.returncode == 0 and len(lines) == 10 and all(ln.endswith("mismatches=0") for ln in lines)
        except (subprocess.SubprocessError, OSError):
            ok = False
    if ok:
        M.BITWISE_VALID, M.REASON = True, "host libm returns the bits elmk_math.h restates (glibc 2.35 FMA algorithms): parity bar = bit identity"
    else:
        M.BITWISE_VALID, M.REASON = False, ("host libm differs from the one elmk_math.h restates: device-vs-oracle tests use the north "
                                            "star's bar (1e-12 relative + per-field floors) instead of bit identity")


def pytest_report_header(config):
    from tests import _parity_mode as M

    return "elmk parity: " + M.REASON


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle is test infrastructure: build it (and oracle/_ref when the reference is mounted) once."""
    from oracle import oracle as O

    O.build(ref=True)
    yield
