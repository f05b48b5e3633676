#!/usr/bin/env python3
"""Turn the SQ counter passes over tests/tools/traffic_probe.py (one rocprofv3 --pmc pass per directory) into the per-kernel
table bench.py's compute_roofline reads.  python tests/tools/make_compute_json.py <out.json> <cols> <tier> <pass dir> [<pass dir> ...]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402  (kernel_source_hash: the table is only valid for the build it was measured on)

out_path, cols, tier = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[4:]:
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("elmk::", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
need = ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
        "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")
doc = {
    "note": f"rocprofv3 --pmc passes (separate runs, counters of one pass in one group), tests/tools/traffic_probe.py, {cols} columns tier {tier}; "
            "per-launch means over the settled launches (the first three launches of every kernel run without scheduling hints and are "
            "left out).  SQ_ACTIVE_INST_VALU is in units of 4 cycles summed over waves, GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
    "source_hash": bench.kernel_source_hash(),
    "columns": cols,
    "tier": tier,
    "kernels": {},
}
for name, ctr in sorted(acc.items()):
    if not name.startswith("k_") or "tile" in name or "transpose" in name or name == "k_copy":
        continue
    if not all(c in ctr for c in need):
        continue
    doc["kernels"][name] = {c: (sum(v[3:]) / len(v[3:]) if len(v) > 5 else sum(v) / len(v)) for c, v in ctr.items()}
    doc["kernels"][name]["launches"] = len(ctr["SQ_INSTS_VALU"])
json.dump(doc, open(out_path, "w"), indent=1)
for name, k in doc["kernels"].items():
    if k["SQ_ACTIVE_INST_VALU"] > 1e6:
        cyc = k["GRBM_GUI_ACTIVE"] / 8
        print(f"{name:28s} busy {k['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024):.2f} lanes {k['SQ_THREAD_CYCLES_VALU'] / (64 * k['SQ_ACTIVE_INST_VALU']):.2f} "
              f"VALU {k['SQ_INSTS_VALU'] / 1e6:7.1f} M  {cyc / 2400:8.1f} us")
