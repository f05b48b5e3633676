#!/usr/bin/env python3
"""GPU box: per-kernel HIP-event times of the 7-kernel step.  python tests/tools/kernel_times.py [cols] [tiers] [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tiers = sys.argv[2] if len(sys.argv) > 2 else "AB"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for tier in tiers:
    D, _ = bench.build_state(cols, 0, tier, 0x5EEDE1A0)
    for _ in range(int(os.environ.get("KT_WARM", "8"))):  # the canopy_fluxes scheduling hints settle over a few steps
        D.restore_fields()
        st.timestep7(D, 1800.0)
    D.restore_fields()
    if os.environ.get("KT_FUSED"):  # the fused step's launch groups instead of the seven wrappers
        for _ in range(4):
            D.restore_fields()
            st.timestep7_fused(D, 1800.0)
        D.restore_fields()
        ms, tot = D.profile_timestep7_fused(1800.0, steps)
        names = st.KERNEL_NAMES_FUSED
    else:
        ms, tot = D.profile_timestep7(1800.0, steps)
        names = st.KERNEL_NAMES
    line = " ".join(f"{n}={m:.3f}" for n, m in zip(names, ms))
    gbs = bench.ALGO_BYTES_STEP * cols / (tot * 1e-3) / 1e9
    print(f"tier {tier} cols {cols}: total {tot:.3f} ms ({cols / tot / 1e3:.1f} M col-steps/s, {gbs:.0f} GB/s algo = {gbs / 80:.1f} % of 8 TB/s) | {line}", flush=True)
    D.close()
