#!/usr/bin/env python3
"""Where the time of one step goes, kernel by kernel, gaps included.
GPU box:   rocprofv3 --kernel-trace -d <dir> -o p --output-format csv -- python3 tests/tools/step_timeline.py run [cols] [tier] [fused]
anywhere:  python3 tests/tools/step_timeline.py parse <dir>     (the last step of the trace: start offset, duration, queue)"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

if sys.argv[1] == "run":
    import bench
    from elmkernels_amd import state as st

    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    tier = sys.argv[3] if len(sys.argv) > 3 else "A"
    fused = len(sys.argv) > 4 and sys.argv[4] == "fused"
    D, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
    adv = st.timestep7_fused if fused else st.timestep7
    if len(sys.argv) > 4 and sys.argv[4] == "advance":
        from elmkernels_amd import synth

        D.set_snow_age_tables(synth.snow_age_tables())
        D.snapshot_fields([k for k in D.fields if k != "err_flags"])
        adv = st.advance_physics
    for _ in range(25):  # (as many steps as bench.py runs: the scheduling hints are a decaying maximum over past steps)
        D.restore_fields()
        D.sync()
        adv(D, 1800.0)
        D.sync()
    print("done")
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps are separated by the restore copies (k_copy): take the kernels after the last k_copy
    last = max(i for i, r in enumerate(rows) if "k_copy" in r["Kernel_Name"])
    step = [r for r in rows[last + 1:] if "elmk::" in r["Kernel_Name"]]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("elmk::", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name}")
        prev_end = max(prev_end, e)
    print(f"step: {(prev_end - t0) / 1e3:.1f} us")
