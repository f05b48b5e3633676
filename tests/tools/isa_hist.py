#!/usr/bin/env python3
"""Opcode-class histogram of one kernel from hipcc's gfx950 assembly (works in the GPU-less container).

  hipcc ... --cuda-device-only -S k_x.hip -o x.s ;  python tests/tools/isa_hist.py x.s <mangled-name-substring> [--top N]

Static counts (every instruction once, whatever its trip count): they say what the code is made of, not what runs.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.endswith("_f64") or "_f64_" in op:
            if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64")):
                return "valu fp64 fma/mul/add"
            if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
                return "valu fp64 trans (rcp/rsq/sqrt)"
            if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
                return "valu fp64 div helpers"
            if op.startswith("v_cmp"):
                return "valu fp64 compare"
            if op.startswith(("v_max_f64", "v_min_f64")):
                return "valu fp64 min/max"
            return "valu fp64 other (" + op + ")"
        if op.startswith("v_cmp"):
            return "valu int/f32 compare"
        if op.startswith("v_cndmask"):
            return "valu cndmask (select)"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "valu mov"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu lane<->sgpr"
        return "valu int/bit/f32"
    if op.startswith("s_"):
        if op.startswith("s_waitcnt") or op.startswith("s_nop"):
            return "s_waitcnt/nop"
        if op.startswith(("s_cbranch", "s_branch")):
            return "salu branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    inside = False
    cls = collections.Counter()
    ops = collections.Counter()
    for line in open(path):
        if not inside:
            if re.match(r"^\S*%s\S*:" % re.escape(name), line):
                inside = True
            continue
        if line.startswith(".Lfunc_end") or line.lstrip().startswith(".section"):
            break
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if not m:
            continue
        op = m.group(1)
        if not op.startswith(("v_", "s_", "ds_", "global_", "flat_", "buffer_", "scratch_")):
            continue
        cls[classify(op)] += 1
        ops[op] += 1
    tot = sum(cls.values())
    valu = sum(v for k, v in cls.items() if k.startswith("valu"))
    print(f"{name}: {tot} instructions, {valu} VALU")
    for k, v in cls.most_common():
        print(f"  {v:7d}  {100.0 * v / tot:5.1f}%  {k}")
    if top:
        print("  top opcodes:")
        for k, v in ops.most_common(top):
            print(f"    {v:7d}  {k}")


if __name__ == "__main__":
    main()
