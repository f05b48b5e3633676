"""Which parity bar the device-vs-oracle tests can hold on THIS host.

The device math library (elmkernels_amd/csrc/elmk_math.h) restates the algorithms of glibc 2.35's x86-64 FMA libm, so that the
GPU returns the bits of the libm the oracle calls: on such a host every fp64 output is compared bit for bit.  On a host
with another libm (or no FMA units) the ORACLE itself differs from a glibc-2.35 reference run in the last bits; the device
is then not wrong, the yardstick moved.  tests/conftest.py detects that once per session (the host build of elmk_math.h
against the live libm on a small sample) and the parity tests fall back to the north star's bar, 1e-12 relative with the
per-field floors of tests/helpers.py.  tests/test_math_host.py stays the loud guard: it fails on such a host."""
BITWISE_VALID = True
REASON = "host libm not checked"
