#!/usr/bin/env python3
"""GPU box: one model step of N columns as TWO blocks - block A (fraction f of the columns) with the leaf-temperature iteration in
256-thread workgroups (ELMK_OPT_CF_HALF_WORKGROUPS), block B with the product's shape - each a context on its own stream, enqueued A
then B and SYNCHRONISED after every step (no overlap across steps: what a single in-order call could do), against one context of N.
GPU_MAX_HW_QUEUES must be above the default 4.   python tests/tools/two_block_stagger.py [N] [tier] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tier = sys.argv[2] if len(sys.argv) > 2 else "A"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sync_each = os.environ.get("STG_SYNC", "1") != "0"


def run(ctxs):
    for _ in range(8):
        for D in ctxs:
            D.restore_fields()
            st.timestep7_fused(D, 1800.0)
    for D in ctxs:
        D.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for D in ctxs:
            D.restore_fields()
            st.timestep7_fused(D, 1800.0)
        if sync_each:
            for D in ctxs:
                D.sync()
    for D in ctxs:
        D.sync()
    return (time.perf_counter() - t0) / steps


D, _ = bench.build_state(n, 0, tier, 0x5EEDE1A0)
t1 = run([D])
D.close()
print(f"one context of {n} columns tier {tier}: {t1 * 1e3:.3f} ms per step (sync after every step: {sync_each})", flush=True)
for f in [float(x) for x in os.environ.get("STG_FRACS", "0.4,0.45,0.5,0.55").split(",")]:
    na = int(n * f) // 256 * 256
    A, _ = bench.build_state(na, 0, tier, 0x5EEDE1A0)
    B, _ = bench.build_state(n - na, 0, tier, 0x5EEDE1A1)
    A.set_option(st.OPT_CF_HALF_WORKGROUPS, 1)
    t = run([A, B])
    print(f"  A = {na} columns (half workgroups) + B = {n - na}: {t * 1e3:.3f} ms per step  ({t / t1 - 1:+.1%})", flush=True)
    A.close()
    B.close()
