#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per launch for every kernel.
python tests/tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if want in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}, "launches", len(next(iter(acc[k].values()))))
