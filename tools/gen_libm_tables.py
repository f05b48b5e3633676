#!/usr/bin/env python3
"""Writes elmkernels_amd/csrc/elmk_math_tables.h: the constant tables of glibc's exp / log / pow / atan (double).

Why: the reference calls the host libm (glibc 2.35 on this image: sysdeps/ieee754/dbl-64/e_exp.c, e_log.c, e_pow.c -
Szabolcs Nagy's table-driven routines, also published as ARM optimized-routines math/exp.c, log.c, pow.c) and the
north-star tolerance is 1e-12 on outputs of an iteration that amplifies a last-bit difference in exp/log/pow.
elmk_math.h restates those three algorithms for the device so that the GPU path returns the same bits as the host libm;
the tables (2^(i/128), 1/c and log(c) for the 128 sub-intervals) are data of that algorithm.  glibc's sources are not in
this image, so the tables are read out of the libm the oracle itself links (the struct is located by its leading
constants, its layout checked against the mathematical definition of every entry).  Run once; the header is committed.

The tables are numerical data of glibc 2.35 routines (exp / log / pow: Szabolcs Nagy, also MIT in Arm optimized-routines;
atan / sincos / asin: IBM Accurate Mathematical Library, LGPL-2.1-or-later): see NOTICE at the repository root.

python3 tools/gen_libm_tables.py [/lib/x86_64-linux-gnu/libm.so.6]
"""
import math
import os
import struct
import subprocess
import sys

LIBM = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
OUT = os.environ.get("ELMK_TABLES_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "elmkernels_amd", "csrc", "elmk_math_tables.h")
blob = open(LIBM, "rb").read()


def d(x):
    return struct.pack("<d", x)


def find_all(pat):
    out, i = [], blob.find(pat)
    while i >= 0:
        out.append(i)
        i = blob.find(pat, i + 1)
    return out


def doubles(off, n):
    return list(struct.unpack_from("<%dd" % n, blob, off))


def u64s(off, n):
    return list(struct.unpack_from("<%dQ" % n, blob, off))


N = 128
# ---- exp: {invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], tab[2N]} -------------------------
(exp_off,) = [o for o in find_all(d(float.fromhex("0x1.71547652b82fep0") * N)) if doubles(o + 8, 1)[0] == float.fromhex("0x1.8p52")]
exp_hdr = doubles(exp_off, 8)
exp_tab = u64s(exp_off + 112, 2 * N)
for i in range(N):  # tab[2i] = bits of the tail of 2^(i/N), tab[2i+1] = bits of 2^(i/N) minus (i << 45)
    tail = struct.unpack("<d", struct.pack("<Q", exp_tab[2 * i]))[0]
    hi = struct.unpack("<d", struct.pack("<Q", (exp_tab[2 * i + 1] + (i << 45)) & (2**64 - 1)))[0]
    assert abs(hi - 2.0 ** (i / N)) <= 2.0**-52 * 2 and abs(tail) < 2.0**-53, i
assert abs(exp_hdr[2] + math.log(2) / N) < 1e-12 and abs(exp_hdr[4] - 0.5) < 1e-9 and abs(exp_hdr[7] - 1 / 120) < 1e-7

# ---- log: {ln2hi, ln2lo, poly[5], poly1[11], tab[N]{invc, logc}} ; pow_log: {ln2hi, ln2lo, poly[7], tab[N]{invc, pad, logc, logctail}}
cands = find_all(d(float.fromhex("0x1.62e42fefa3800p-1")))
log_off = pow_off = None
for o in cands:
    t = doubles(o + 144, 2 * N)
    if all(abs(t[2 * i + 1] + math.log(t[2 * i])) < 1e-12 and 0.5 < t[2 * i] < 1.6 for i in range(N) if t[2 * i] > 0) and t[0] > 0:
        log_off = o
    t = doubles(o + 72, 4 * N)
    if all(t[4 * i] > 0 and t[4 * i + 1] == 0.0 and abs(t[4 * i + 2] + t[4 * i + 3] + math.log(t[4 * i])) < 1e-15 for i in range(N)):
        pow_off = o
assert log_off is not None and pow_off is not None and log_off != pow_off
log_hdr = doubles(log_off, 18)
log_tab = doubles(log_off + 144, 2 * N)
pow_hdr = doubles(pow_off, 9)
pow_tab = doubles(pow_off + 72, 4 * N)
assert abs(log_hdr[2] + 0.5) < 1e-15 and abs(log_hdr[3] - 1 / 3) < 1e-9 and log_hdr[7] == -0.5 and abs(log_hdr[8] - 1 / 3) < 1e-12
assert pow_hdr[2] == -0.5 and abs(pow_hdr[3] + 2 / 3) < 1e-9  # pow's polynomial is in powers of (-2 r): A[1] = (1/3)·(-2)

# ---- atan (IBM Accurate Mathematical Library, sysdeps/ieee754/dbl-64/s_atan.c + uatan.tbl): cij[241][7], row i =
# {x_i, atan(x_i), c2..c6} for x_i ~ (i + 16) / 256; located by the first entry of the table
atan_offs = find_all(d(float.fromhex("0x1.0400665e0244ep-4")) + d(float.fromhex("0x1.03a737b53dd20p-4")))
atan_off = atan_offs[0]
atan_tab = doubles(atan_off, 7 * 241)
assert all(doubles(o, 7 * 241) == atan_tab for o in atan_offs)  # one copy per ifunc variant (sse2 / fma / fma4), all equal
for i in range(241):
    assert abs(atan_tab[7 * i] - (i + 16) / 256) < 1 / 300 and abs(atan_tab[7 * i + 1] - math.atan(atan_tab[7 * i])) < 1e-13, i
    assert abs(atan_tab[7 * i + 2] - 1 / (1 + atan_tab[7 * i] ** 2)) < 1e-12, i  # c2 = atan'(x_i)

# ---- sin / cos (IBM Accurate Mathematical Library, sysdeps/ieee754/dbl-64/s_sin.c + sincostab.c): __sincostab[440],
# entry k = {sin(x_k) high, low, cos(x_k) high, low} for x_k = k / 128; located by its first non-trivial entry
sc_offs = find_all(d(0.0) + d(0.0) + d(1.0) + d(0.0) + d(float.fromhex("0x1.fffeaaaaeeeefp-8")))
sincos_tab = doubles(sc_offs[0], 440)
assert all(doubles(o, 440) == sincos_tab for o in sc_offs)
for k in range(110):
    assert abs(sincos_tab[4 * k] + sincos_tab[4 * k + 1] - math.sin(k / 128)) < 1e-15 and abs(sincos_tab[4 * k + 2] - math.cos(k / 128)) < 1e-15, k

# ---- asin / acos (IBM Accurate Mathematical Library, sysdeps/ieee754/dbl-64/e_asin.c + asincos.tbl, root.tbl): asncs[2568]
# (rows {x_i, c1.., asin(x_i) high, low} of five different lengths for the ranges of |x| in [0.125, 0.96875)) and
# inroot[128] (1/sqrt seeds); located by their first entries
as_offs = find_all(d(float.fromhex("0x1.04p-3")) + d(float.fromhex("0x1.0216988994424p+0")))
asncs_tab = doubles(as_offs[0], 2568)
assert all(doubles(o, 2568) == asncs_tab for o in as_offs)
assert abs(asncs_tab[8] - math.asin(asncs_tab[0])) < 1e-15 and abs(asncs_tab[1] - 1 / math.sqrt(1 - asncs_tab[0] ** 2)) < 1e-15
ir_offs = find_all(d(float.fromhex("0x1.68a1f80d7182p+0")) + d(float.fromhex("0x1.65de82af9631fp+0")))
inroot_tab = doubles(ir_offs[0], 128)
assert all(doubles(o, 128) == inroot_tab for o in ir_offs)
for i in range(128):  # 1/sqrt of the midpoint of the i-th of 128 intervals of [0.5, 2)
    assert 0.70 < inroot_tab[i] < 1.42, i

ver = subprocess.run(["ldd", "--version"], capture_output=True, text=True).stdout.splitlines()[0]


def hx(x):
    return "0x%016xull" % struct.unpack("<Q", struct.pack("<d", x))[0]


def emit(name, vals, per=4, conv=hx):
    s = "ELMK_MATH_TABLE %s[%d] = {\n" % (name, len(vals))
    for i in range(0, len(vals), per):
        s += "  " + ", ".join(conv(v) for v in vals[i : i + per]) + ",\n"
    return s + "};\n"


with open(OUT, "w") as f:
    f.write("// GENERATED by tools/gen_libm_tables.py - do not edit.  Constant tables (IEEE-754 binary64 bit patterns) of the\n")
    f.write("// exp / log / pow of %s, read from its libm.so.6.\n" % ver)
    f.write("// Layout: the __exp_data / __log_data / __pow_log_data structs of sysdeps/ieee754/dbl-64/math_config.h.\n")
    f.write("// Numerical data of glibc routines: exp / log / pow tables by Szabolcs Nagy (also MIT in Arm optimized-routines); atan /\n")
    f.write("// sincos / asin tables of the IBM Accurate Mathematical Library (LGPL-2.1-or-later).  See NOTICE at the repository root.\n")
    f.write("#pragma once\n#include <stdint.h>\n\n")
    f.write("// invln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5\n")
    f.write(emit("elmk_exp_hdr", exp_hdr))
    f.write("// 2^(i/128): [2i] = tail bits, [2i+1] = bits minus (i << 45)\n")
    f.write(emit("elmk_exp_tab", exp_tab, conv=lambda v: "0x%016xull" % v))
    f.write("// ln2hi, ln2lo, A[0..4], B[0..10]\n")
    f.write(emit("elmk_log_hdr", log_hdr))
    f.write("// {1/c, log(c)} for the 128 sub-intervals of [0x1.6p-1, 0x1.6p0)\n")
    f.write(emit("elmk_log_tab", log_tab))
    f.write("// ln2hi, ln2lo, A[0..6]\n")
    f.write(emit("elmk_powlog_hdr", pow_hdr))
    f.write("// {1/c, log(c) high part, log(c) tail}\n")
    f.write(emit("elmk_powlog_tab", [pow_tab[4 * i + j] for i in range(N) for j in (0, 2, 3)], per=3))
    f.write("// s_atan.c cij[241][7]: {x_i, atan(x_i), c2..c6}, x_i ~ (i + 16) / 256\n")
    f.write(emit("elmk_atan_tab", atan_tab, per=7))
    f.write("// sincostab.c __sincostab[440]: {sin hi, sin lo, cos hi, cos lo} of k / 128\n")
    f.write(emit("elmk_sincos_tab", sincos_tab, per=4))
    f.write("// asincos.tbl asncs[2568]\n")
    f.write(emit("elmk_asncs_tab", asncs_tab, per=4))
    f.write("// root.tbl inroot[128]\n")
    f.write(emit("elmk_inroot_tab", inroot_tab, per=4))
print("wrote", os.path.normpath(OUT), "exp@%#x log@%#x pow_log@%#x" % (exp_off, log_off, pow_off))
