#!/usr/bin/env python3
"""GPU box, development: trip-count histogram of canopy_fluxes and, with a CF_PROBE=4 build selected through
ELMK_LIBRARY, the per-wave timeline of k_cf_iterate.  python tests/tools/cf_timeline.py [cols]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from elmkernels_amd import state as st  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, _ = bench.build_state(cols, 0, os.environ.get("CF_TIER", "B"), 0x5EEDE1A0)
for _ in range(int(os.environ.get('CF_WARM', '2'))):
    D.restore_fields()
    st.timestep7(D, 1800.0)
prev = D.canopy_trip_counts()
hint = D.canopy_schedule_hints()
D.restore_fields()
st.timestep7(D, 1800.0)
D.sync()
trips = D.canopy_trip_counts()
d = trips - prev
print("trip count change between the last two steps: changed", int((d != 0).sum()), "|d|>5:", int((abs(d) > 5).sum()),
      "hint<=5 but now >=12:", int(((prev <= 5) & (prev > 0) & (trips >= 12)).sum()), "max", int(abs(d).max()))
veg = trips > 0
longc = trips >= 30
print("columns with >= 30 trips now:", int(longc.sum()), "of which the scheduler's hint was < 22:", int((longc & (hint < 22)).sum()),
      "< 16:", int((longc & (hint < 16)).sum()), "< 12:", int((longc & (hint < 12)).sum()))
day = (D.download("nrad") > 0) & ((D.download("parsun_z").reshape(cols, -1)[:, 0] > 0) | (D.download("parsha_z").reshape(cols, -1)[:, 0] > 0))
for lo, hi in ((1, 5), (6, 8), (9, 11), (12, 15), (16, 21), (22, 41)):
    for nm, mk in (("day", day), ("night", ~day)):
        m = (prev >= lo) & (prev <= hi) & mk
        print(f"prev {lo}-{hi} {nm}: n {int(m.sum())}  now>=12: {int((trips[m] >= 12).sum())}  now>=20: {int((trips[m] >= 20).sum())}  now>=30: {int((trips[m] >= 30).sum())}")
print(f"columns {cols}: vegetated {veg.sum()} day {int((veg & day).sum())} night {int((veg & ~day).sum())}")
for name, m in (("day", veg & day), ("night", veg & ~day)):
    t = trips[m]
    h = np.bincount(t, minlength=42)
    print(name, "mean trips %.2f" % t.mean(), "hist", {i: int(v) for i, v in enumerate(h) if v})
ld = D.level_stride
WK_DEBUG = 1
nw = 4096
w = D.read_work(WK_DEBUG * ld, nw * 32).reshape(nw, 32)
w = w[w[:, 7] == 1.0]
if len(w):
    t0 = w[:, 0].min()
    start, exh, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0  # us
    exh = np.where(w[:, 1] > 0, exh, np.nan)
    print(f"waves that ran: {len(w)}; start max {start.max():.0f} us; queue exhausted at {np.nanmin(exh):.0f}..{np.nanmax(exh):.0f} us")
    print("end time percentiles (us):", {p: round(float(np.percentile(end, p))) for p in (0, 10, 50, 90, 99, 100)})
    busy = w[:, 3] > 0
    print(f"waves with work: {busy.sum()}, trips/wave mean {w[busy, 3].mean():.1f} max {w[busy, 3].max():.0f}; "
          f"lane utilisation over trips {w[busy, 4].sum() / (64 * w[busy, 3].sum()):.3f}; refills/wave {w[busy, 5].mean():.1f}; "
          f"us per trip {((end - start)[busy].sum() / w[busy, 3].sum()):.1f}")
    late = np.argsort(end)[-5:]
    print("latest waves: end us", np.round(end[late]), "trips", w[late, 3], "cols", w[late, 6], "refills", w[late, 5])
    print(f"tail: last wave ends {end.max() - np.nanmin(exh):.0f} us after the queue ran dry (kernel {end.max():.0f} us)")
    print(f"lane-trips {w[busy, 4].sum():.0f}: day {w[busy, 12].sum() / w[busy, 4].sum():.3f}; Brent used in {w[busy, 15].sum() / w[busy, 4].sum():.4f} of lane-trips, "
          f"{w[busy, 14].sum() / w[busy, 3].sum():.3f} of wave-trips; wave-trips with a C4 lane {w[busy, 13].sum() / w[busy, 3].sum():.3f}; "
          f"wave-trips with a day lane {w[busy, 8].sum() / w[busy, 3].sum():.3f}")
    # root-find statistics: ci_func evaluations per lane (what each column needs) against ci_func bodies the wave executed
    for ph, name in ((0, "sunlit"), (1, "shaded")):
        lane_ev, wave_ev = w[busy, 28 + ph].sum(), w[busy, 30 + ph].sum()
        if wave_ev > 0:
            print(f"ci_func {name}: {lane_ev:.0f} lane evaluations, {wave_ev:.0f} executed wave bodies -> active-lane fraction "
                  f"{lane_ev / (64 * wave_ev):.3f}; bodies per day wave-trip {wave_ev / w[busy, 8].sum():.2f}; "
                  f"evaluations per day lane-trip {lane_ev / w[busy, 12].sum():.2f}")
    sec = w[busy, 16:28].sum(axis=0)
    if sec.sum() > 0:
        names = ["refill+record load", "friction profiles", "resistances", "psn_temp + phase inputs", "solve sunlit", "solve shaded",
                 "energy balance", "qsat + Monin-Obukhov + stop test", "store fin", "-", "-", "-"]
        print("shader-clock share per section:", {n: round(float(v / sec.sum()), 3) for n, v in zip(names, sec) if n != "-"},
              "cycles/trip", round(float(sec.sum() / w[busy, 3].sum())))
else:
    print("no timeline (product build)")
D.close()
