#!/usr/bin/env python3
"""Registers, spills, LDS and the occupancy they allow, for every kernel - from hipcc's gfx950 assembly, in the GPU-less
build container.

  python tests/tools/kernel_resources.py <file.s | k_x.hip> [name-filter] [extra hipcc flags]     one file, as a table
  python tests/tools/kernel_resources.py --json profiles/r03_kernel_resources.json             every kernel file, product build
                                                                                                 (fp64 state) and -DELMK_STATE_F32

waves/SIMD = min(8, 512 / VGPRs rounded up to 8, LDS: 160 KiB per CU / workgroup LDS x waves per workgroup / 4 SIMDs)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CSRC = os.path.join(ROOT, "elmkernels_amd", "csrc")
KFILES = ["k_water_energy", "k_canopy_fluxes", "k_albedo_snicar", "k_soil_temperature", "k_snow_hydrology", "k_surface_fluxes", "k_forcing",
          "k_init_state", "k_util"]


def compile_s(src, extra=()):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-mllvm", "-disable-machine-licm", f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}", "--cuda-device-only", "-S", src,
                           "-o", out] + list(extra), stderr=subprocess.DEVNULL)
    return out


def parse(path):
    txt = open(path).read()
    res = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        name = g("name")
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        short = short.replace("void ", "").replace("elmk::", "")
        vgpr, lds, wg = int(g("vgpr_count")), int(g("group_segment_fixed_size")), int(g("max_flat_workgroup_size"))
        by_vgpr = min(8, 512 // max(8, (vgpr + 7) // 8 * 8))
        wg_waves = max(1, wg // 64)
        by_lds = 8 if lds == 0 else min(8, (163840 // lds) * wg_waves // 4)
        res[short] = {"vgpr": vgpr, "vgpr_spill": int(g("vgpr_spill_count")), "sgpr": int(g("sgpr_count")), "sgpr_spill": int(g("sgpr_spill_count")),
                      "lds_bytes": lds, "scratch_bytes": int(g("private_segment_fixed_size")), "workgroup": wg,
                      "waves_per_simd": max(0, min(by_vgpr, by_lds)), "limited_by": "lds" if by_lds < by_vgpr else "vgpr"}
    return res


def main():
    if sys.argv[1] == "--json":
        sys.path.insert(0, ROOT)
        import bench

        doc = {"note": "hipcc --offload-arch=gfx950 metadata of every kernel, product build (fp64 state) and -DELMK_STATE_F32 (fp64 fields stored as "
                       "fp32); waves_per_simd = min(8, 512 / VGPRs, LDS limit)", "source_hash": bench.kernel_source_hash(), "kernels": {}}
        # the per-translation-unit flags of the product build (FLAGS_<file> in elmkernels_amd/csrc/Makefile: nontemporal state accesses)
        unit_flags = {}
        for line in open(os.path.join(CSRC, "Makefile")):
            m = re.match(r"FLAGS_(\w+)\s*:=\s*(.*)$", line)
            if m:
                unit_flags[m.group(1)] = m.group(2).split()
        doc["unit_flags"] = {k: " ".join(v) for k, v in unit_flags.items()}
        for f in KFILES:
            a = parse(compile_s(os.path.join(CSRC, f + ".hip"), unit_flags.get(f, [])))
            b = parse(compile_s(os.path.join(CSRC, f + ".hip"), ["-DELMK_STATE_F32"] + unit_flags.get(f, [])))
            for k in a:
                doc["kernels"][k] = {"f64_state": a[k], "f32_state": b.get(k)}
        json.dump(doc, open(sys.argv[2], "w"), indent=1)
        for k, v in doc["kernels"].items():
            a, b = v["f64_state"], v["f32_state"] or {}
            print(f"{k[:34]:36s} fp64 state: vgpr {a['vgpr']:3d} lds {a['lds_bytes']:6d} -> {a['waves_per_simd']} waves/SIMD | fp32 state: vgpr "
                  f"{b.get('vgpr', 0):3d} -> {b.get('waves_per_simd', 0)} waves/SIMD")
        return
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    if path.endswith(".hip"):
        path = compile_s(path if os.path.exists(path) else os.path.join(CSRC, path), sys.argv[3:])
    for k, v in parse(path).items():
        if flt in k:
            print(f"{k[:44]:46s} vgpr {v['vgpr']:4d} (spill {v['vgpr_spill']})  sgpr {v['sgpr']:4d} (spill {v['sgpr_spill']})  lds {v['lds_bytes']:7d}  "
                  f"scratch {v['scratch_bytes']}  -> {v['waves_per_simd']} waves/SIMD ({v['limited_by']})")


if __name__ == "__main__":
    main()
