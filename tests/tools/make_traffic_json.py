#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tests/tools/traffic_probe.py into the per-kernel
HBM byte table bench.py reads.  python tests/tools/make_traffic_json.py <fetch dir> <write dir> <out.json> [cols]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402  (kernel_source_hash: the table is only valid for the build it was measured on)


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name].append(float(row["Counter_Value"]))
    # k_copy is launched with two sizes (the 1 GiB calibration copies and the small snapshot restores): keep the calibration
    return {k: (max(v) if k == "elmk::k_copy" else sum(v) / len(v)) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
tier = sys.argv[5] if len(sys.argv) > 5 else "B"
out = {
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), tests/tools/traffic_probe.py, "
            f"{cols} columns tier {tier}; values are per-launch means in KiB as reported. Calibration on elmk::k_copy (1 GiB read + "
            "1 GiB written with the kernels' own 8-byte-per-lane access shape): FETCH_SIZE reads exactly 1/2 of the bytes, "
            "WRITE_SIZE reads them exactly (as MI355X_MICROARCH.md says for wide streams) -> "
            "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.",
    "source_hash": bench.kernel_source_hash(),
    "columns": cols,
    "tier": tier,
    "kernels": {},
}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("elmk::k_") or "tile" in k or "transpose" in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                         "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
c = out["kernels"].get("elmk::k_copy")
if c:
    print("calibration k_copy: FETCH KiB", c["FETCH_SIZE_KiB"], "WRITE KiB", c["WRITE_SIZE_KiB"], "(1 GiB = 1048576 KiB each way)")
tot = sum(v["hbm_bytes_per_launch"] for k, v in out["kernels"].items() if k != "elmk::k_copy")
print("HBM bytes per timestep (all physics kernels):", tot, "=", round(tot / cols), "B/column")
