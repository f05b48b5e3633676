#!/usr/bin/env python3
"""GPU box: does the fp64-bound leaf-temperature iteration of one block of columns overlap with the streaming kernels of
another?  Two contexts of N columns each (own arena, own stream) stepped alternately from one host thread, against one context of
2 N columns; ELMK_CF_GROUPS limits the persistent k_cf_iterate grid so that it leaves CUs free (it holds a CU's whole LDS).
python tests/tools/two_ctx_overlap.py [N] [tier] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tier = sys.argv[2] if len(sys.argv) > 2 else "A"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = os.environ.get("OVL_MODE", "two")


def run(ctxs, steps):
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
    for D in ctxs:
        D.sync()
    return (time.perf_counter() - t0) / steps


if mode == "one":
    D, _ = bench.build_state(2 * n, 0, tier, 0x5EEDE1A0)
    t = run([D], steps)
    print(f"one context of {2 * n} columns tier {tier}: {t * 1e3:.3f} ms per step, {2 * n / t / 1e6:.1f} M gridcell-steps/s (ELMK_CF_GROUPS={os.environ.get('ELMK_CF_GROUPS', '-')})")
else:
    k = int(os.environ.get("OVL_BLOCKS", "2"))  # the 2 n columns as k contexts
    m = 2 * n // k
    ctxs = [bench.build_state(m, 0, tier, 0x5EEDE1A0 + i)[0] for i in range(k)]
    t = run(ctxs, steps)
    print(f"{k} contexts of {m} columns tier {tier}: {t * 1e3:.3f} ms per step of all, {k * m / t / 1e6:.1f} M gridcell-steps/s (ELMK_CF_GROUPS={os.environ.get('ELMK_CF_GROUPS', '-')})")
