#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in hipcc's gfx950 assembly (.s from --cuda-device-only -S), or of the kernels of
one .hip file (compiled here with the product flags).  python tests/tools/kernel_resources.py <file.s | k_x.hip> [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
if path.endswith(".hip"):
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    csrc = os.path.join(root, "elmkernels_amd", "csrc")
    src = path if os.path.exists(path) else os.path.join(csrc, path)
    out = tempfile.mktemp(suffix=".s")
    extra = sys.argv[3:] if len(sys.argv) > 3 else []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-mllvm", "-disable-machine-licm", f"-I{csrc}", f"-I{os.path.join(root, 'include')}", "--cuda-device-only", "-S", src,
                           "-o", out] + extra, stderr=subprocess.DEVNULL)
    path = out
txt = open(path).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    short = re.sub(r"^_ZN4elmk\d+", "", name)
    if flt in name:
        print(f"{short[:44]:46s} vgpr {g('vgpr_count'):>4s} (spill {g('vgpr_spill_count')})  sgpr {g('sgpr_count'):>4s} (spill {g('sgpr_spill_count')})  "
              f"lds {g('group_segment_fixed_size'):>7s}  scratch {g('private_segment_fixed_size')}")
