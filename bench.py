#!/usr/bin/env python3
"""bench.py - gridcell-timesteps/sec of the full water+energy timestep (all 7 kernels) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cols C] [--tier A|B] [--workload timestep7|soil_temperature] [--fused]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  `python bench.py --gpus N` without a launcher starts the N rank processes itself (the parent never
touches the GPU and never re-executes: it only spawns children, relays rank 0's line and returns the worst exit code);
under torch.distributed.run the ranks come from RANK / LOCAL_RANK / WORLD_SIZE.  BASELINE config 4 (80 M columns over
8 GPUs) is `--gpus 8 --cols 10000000`.

Columns shard halo-free across ranks (elmkernels_amd.decomp: the reference's 1-D block split, src/utils/utils.cc:27-44);
there is no data-path collective - torch.distributed (RCCL) only carries the barrier and the max-over-ranks of the timed
region.  Per-GPU work is fixed (weak scaling): --cols columns on every rank.

A "step" = one pass of the reference's ELMInterface::advance hot path over the resident state: restore of the
snapshot fields (t_veg + forcing heights: what the rest of the model does between steps, 64 B/column), then
frac_wet -> albedo_snicar -> canopy_hydrology -> surface_radiation -> canopy_temperature -> bareground_fluxes
-> canopy_fluxes, each a hand-written HIP kernel over SoA state in HBM (--fused: the same step through
elmk_timestep7_fused, one streaming stage for the wrappers between albedo and the leaf-temperature iteration).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timed on the launch stream), `cpu_baseline` (the C oracle and, where oracle/_ref was built, the reference's own headers,
both with OpenMP on the host cores, bounded sample) and, at N = 1 on the default workload, `north_star_10M` (the same
step at the north-star size, 10 M columns, both synthetic tiers).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per column-step (SURVEY.md section 8(d) / BASELINE.md section 3; DESIGN.md restates the tally)
ALGO_BYTES = {
    "frac_wet": 44, "albedo_snicar": 960, "canopy_hydrology": 440, "surface_radiation": 640,
    "canopy_temperature": 753, "bareground_fluxes": 368, "canopy_fluxes": 2076,
}
ALGO_BYTES_STEP = sum(ALGO_BYTES.values())  # 5281
ALGO_BYTES_FUSED = 3405  # fused lower bound: every state element read once + written once per step
# what a predicate-gated wrapper touches on a column it does NOT work on: albedo's init_timestep defaults (500 B written),
# bareground's cgrnd* reset (24 B), canopy_fluxes' bare branch (32 B)
GATED_IDLE_BYTES = {"albedo_snicar": 500, "bareground_fluxes": 24, "canopy_fluxes": 32}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
RESTORE_FIELDS = ["t_veg", "forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch"]
NBASE = 47 * 64 * 8  # host-generated base block; the device tiles it to --cols
NORTH_STAR_COLS = 10_000_000
TIER_NAMES = {"A": "fixture-tiled", "B": "branch-mix"}

# Test hook (tests/test_sharding_gloo.py): a factory that stands in for build_state so that the multi-rank control flow
# of this file - rank discovery, self-spawn, block split, barrier, max-over-ranks, rank-0 aggregation, the JSON line -
# runs on CPU under gloo.  "module:function", imported only when the variable is set; the product path never sets it.
REHEARSAL_ENV = "ELMK_BENCH_REHEARSAL"


def build_state(ncols, device, tier, seed, lib_path=None):
    from elmkernels_amd import state as st
    from elmkernels_amd import synth

    ft = st.field_table()
    nbase = min(NBASE, ncols)
    cols, scal, soil = synth.make_state(ft, nbase, tier=tier, seed=seed)
    D = st.ELMState(ncols, device, lib_path=lib_path)
    pft, optics = synth.load_params()
    D.set_pft(pft)
    D.set_snicar(optics)
    D.set_soilcolor(soil["albsat"], soil["albdry"])
    D.set_land(**synth.TEST_LAND)
    D.set_scalars(**scal)
    for k, v in cols.items():
        D.upload(k, v, col0=0)
    if ncols > nbase:
        D.tile_columns(nbase, seed=seed, rules=synth.TILE_RULES)
    D.snapshot_fields(RESTORE_FIELDS)
    D.sync()
    return D, (cols, scal, soil)


def _timed_loop(fn, budget_s, min_steps=3, max_steps=100000):
    steps = 0
    t0 = time.perf_counter()
    while True:
        fn()
        steps += 1
        el = time.perf_counter() - t0
        if (el > budget_s and steps >= min_steps) or steps >= max_steps:
            return steps, el


def host_cpu_budget():
    """Hardware threads this process may really use: the smaller of its affinity mask and its cgroup CPU quota.
    -> (threads, {"affinity": .., "cgroup_quota_cpus": .. or None, "nproc": ..})"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cap = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cap, {"affinity": aff, "cgroup_quota_cpus": quota, "nproc": os.cpu_count()}


# read at import, before libgomp is loaded: with OMP_PROC_BIND set libgomp binds the calling thread to ONE place of the mask, and
# the mask read afterwards would say 1
HOST_CPU_BUDGET = host_cpu_budget()
CPU_COLS_PER_THREAD = 2048  # >= 2 000 columns per OpenMP thread in every timing


def cpu_baseline(host_state, budget_s=12.0, workload="timestep7"):
    """The CPU path beside the GPU numbers, on a bounded sample of the same workload.
      * kind "reference": the seven wrappers run by the reference's OWN physics headers (oracle/_ref/libelmref.so +
        libelmref_canopy.so, compiled in the build container) under `#pragma omp parallel for` over columns - the execution
        shape of Kokkos::parallel_for(RangePolicy<OpenMP>) (src/utils/invoke_kernel.hh:40-46);
      * "port" (beside it; the only figure for the soil-temperature workload or when oracle/_ref is absent): the oracle
        (plain-C restatement of the reference physics, OpenMP over columns).
    The thread count is swept (1, 8, 32, all the process may use: affinity mask and cgroup quota, not omp_get_max_threads of a
    shared host), every timing on >= CPU_COLS_PER_THREAD columns per thread (the host-generated base block tiled on the host),
    threads bound close and waiting passively (set in main() before libgomp loads); `value` is the best of the sweep, the sweep
    and the parallel efficiency of the best point (value / (threads x the 1-thread rate)) are beside it."""
    import numpy as np

    from tests import helpers as H

    cols, scal, soil = host_state
    from oracle import oracle as O

    cap, hostinfo = HOST_CPU_BUDGET
    omp_max = int(O.lib().lib.elmo_get_max_threads())
    cap = max(1, min(cap, omp_max))
    nb = next(iter(cols.values())).shape[0]
    nbase = min(nb, NBASE)
    use_ref = workload == "timestep7" and O.have_ref() and O.have_ref_canopy()
    R = O.Reference() if use_ref else None

    def set_threads(t):
        O.lib().lib.elmo_set_threads(int(t))
        if R is not None:
            if hasattr(R.R, "elmref_set_threads"):
                R.R.elmref_set_threads(int(t))
            rc = getattr(O.lib(), "ref_canopy", None)
            if rc is not None and hasattr(rc, "elmref_canopy_set_threads"):
                rc.elmref_canopy_set_threads(int(t))

    def state_of(n):
        reps = -(-n // nbase)
        sub = {k: (np.concatenate([v[:nbase]] * reps)[:n] if reps > 1 else v[:n]) for k, v in cols.items()}
        return H.oracle_state(sub, scal, soil)

    def stepper(S, ref):
        tveg = S["t_veg"].copy()
        hg = {k: S[k].copy() for k in RESTORE_FIELDS[1:]}
        if workload == "soil_temperature":
            S.timestep7(1800.0)
            saved = {k: S[k].copy() for k in SOIL_RESTORE}

            def step():
                for k, v in saved.items():
                    S[k][...] = v
                S.soil_temperature(1800.0)
        elif ref:
            def step():
                S["t_veg"][:] = tveg
                for k, v in hg.items():
                    S[k][:] = v
                R.frac_wet(S)
                S.albedo_snicar_ref()
                R.canopy_hydrology(S, 1800.0)
                R.surface_radiation(S)
                R.canopy_temperature(S)
                R.bareground_fluxes(S)
                S.canopy_fluxes_ref(1800.0)
        else:
            def step():
                S["t_veg"][:] = tveg
                for k, v in hg.items():
                    S[k][:] = v
                S.timestep7(1800.0)
        return step

    sweep_t = sorted({t for t in (1, 8, 32, cap) if t <= cap})
    per_point = budget_s / (len(sweep_t) + 1)
    sweep = []
    for t in sweep_t:
        set_threads(t)
        n = max(CPU_COLS_PER_THREAD * t, min(nbase, 4096))
        S = state_of(n)
        step = stepper(S, use_ref)
        step()  # warm-up (thread pool, page faults)
        steps, el = _timed_loop(step, per_point, min_steps=2)
        sweep.append({"threads": t, "columns": n, "steps": steps, "seconds": round(el, 2), "value": n * steps / el})
        del S
    best = max(sweep, key=lambda r: r["value"])
    one = next(r for r in sweep if r["threads"] == 1)
    kind = "reference" if use_ref else "port"
    what = ("all seven wrappers by the reference's own physics headers (oracle/_ref/libelmref.so + libelmref_canopy.so, g++ -O2 "
            "-fopenmp, omp parallel for over columns)") if use_ref else "oracle/libelmoracle.so (gcc -O2 -fopenmp)"
    unit = "columns/s" if workload == "soil_temperature" else "gridcell-timesteps/s"
    out = {
        "value": best["value"], "unit": unit, "cores": best["threads"], "kind": kind,
        "sample": f"{best['columns']} columns x {best['steps']} timesteps of the same tier ({CPU_COLS_PER_THREAD} columns per thread, the "
                  f"{nbase}-column base block tiled), {what}, {best['seconds']:.1f} s; OMP_PROC_BIND=close, OMP_WAIT_POLICY=passive",
        "sweep": sweep,
        "parallel_efficiency": best["value"] / (best["threads"] * one["value"]),
        "one_thread_us_per_column_step": 1e6 / one["value"],
        "host": dict(hostinfo, omp_max_threads=omp_max, threads_usable=cap),
    }
    if use_ref:  # the port beside it, at the best thread count
        set_threads(best["threads"])
        S = state_of(best["columns"])
        step = stepper(S, False)
        step()
        steps, el = _timed_loop(step, per_point, min_steps=2)
        out["port"] = {"value": best["columns"] * steps / el, "unit": unit, "cores": best["threads"], "kind": "port",
                       "sample": f"{best['columns']} columns x {steps} timesteps, oracle/libelmoracle.so (gcc -O2 -fopenmp), {el:.1f} s"}
    set_threads(cap)
    return out


# HIP kernels behind each wrapper / launch group: name prefixes as rocprofv3 reports them
KERNEL_PREFIX = {
    "frac_wet": ("elmk::k_frac_wet",),
    "albedo_snicar": ("elmk::k_alb_",),
    "canopy_hydrology": ("elmk::k_canopy_hydrology",),
    "surface_radiation": ("elmk::k_surface_radiation",),
    "canopy_temperature": ("elmk::k_canopy_temperature",),
    "bareground_fluxes": ("elmk::k_bg_",),
    "canopy_fluxes": ("elmk::k_cf_",),
    "soil_temperature": ("elmk::k_st_", "elmk::k_soil_temperature"),
    "fused_stream": ("elmk::k_fz_",),
    "canopy_iterate": ("elmk::k_cf_iterate", "elmk::k_cf_finish"),
}
PROFILE_TAG = "r04"


def kernel_source_hash():
    """Hash of the kernel sources: the committed PMC tables carry the hash of the build they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "elmkernels_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) and name != "elmk_math_tables.h":
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(group, tier, cols, table=None):
    """HBM bytes per launch of the group's kernels from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    calibrated on gfx950, WRITE_SIZE as is; profiles/<round>_hbm_traffic_pmc_tier{A,B}.json).  The counters cannot be
    read from inside this process, so the number is only reported for the configuration and the build (source hash) it
    was measured on; otherwise None, with the reason on stderr."""
    path = os.path.join(ROOT, "profiles", table or f"{PROFILE_TAG}_hbm_traffic_pmc_tier{tier}.json")
    if not os.path.exists(path) or group not in KERNEL_PREFIX:
        return None
    doc = json.load(open(path))
    if int(doc.get("columns", 1_000_000)) != cols:
        return None
    if doc.get("source_hash") not in (None, kernel_source_hash()):
        print(f"bench.py: {os.path.basename(path)} was measured on another build of the kernels: roofline.traffic = null",
              file=sys.stderr)
        return None
    k = doc["kernels"]
    names = [n for n in k if n.startswith(KERNEL_PREFIX[group])]
    if not names:
        print(f"bench.py: no kernel of {group} in {os.path.basename(path)}: roofline.traffic = null", file=sys.stderr)
        return None
    return float(sum(k[n]["hbm_bytes_per_launch"] for n in names))


FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector fp64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (vendor figure)
N_SIMD = 1024


def compute_roofline(kernel_names, tier, cols, table=None):
    """The compute side of the roofline for kernels that are not bandwidth-bound, from the committed SQ counter passes
    (profiles/<round>_compute_pmc_tier{A,B}.json, tests/tools/make_compute_json.py; separate rocprofv3 --pmc passes).  Like
    `traffic` it is only reported for the configuration and the build (source hash) it was measured on.
      valu_busy       = SQ_ACTIVE_INST_VALU * 4 / (kernel cycles * 1024 SIMDs): share of the cycles a SIMD issues VALU work
      valu_lane_util  = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): share of the 64 lanes doing it (divergence)
      fp64_tflops     = (2 FMA + MUL + ADD + TRANS wave-instructions) * 64 lanes / kernel time: the issue rate of fp64
                        arithmetic, all 64 lanes counted; x valu_lane_util = what the columns actually got"""
    path = os.path.join(ROOT, "profiles", table or f"{PROFILE_TAG}_compute_pmc_tier{tier}.json")
    if cols != 1_000_000 or not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if doc.get("source_hash") not in (None, kernel_source_hash()):
        print(f"bench.py: {os.path.basename(path)} was measured on another build of the kernels: compute_roofline = null", file=sys.stderr)
        return None
    out = {}
    for name in kernel_names:
        k = doc["kernels"].get(name)
        if not k:
            continue
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0
        secs = cyc / 2.4e9
        busy = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD)
        lane = k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"])
        flops = (2 * k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_TRANS_F64"]) * 64.0
        tf = flops / secs / 1e12
        out[name] = {"valu_busy": round(busy, 3), "valu_lane_util": round(lane, 3), "fp64_tflops": round(tf, 2),
                     "frac_of_fp64_peak": round(tf / FP64_VECTOR_PEAK_TFLOPS, 3), "fp64_tflops_active_lanes": round(tf * lane, 2),
                     "valu_insts_per_launch": k["SQ_INSTS_VALU"], "fp64_share_of_valu_insts": round(
                         (k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_TRANS_F64"])
                         / k["SQ_INSTS_VALU"], 3), "avg_launch_ms_counter_pass": round(secs * 1e3, 4)}
    return out or None


# SIMD time per wave64 instruction at two or more waves per SIMD, measured on MI355X (tests/tools/ubench/valu_cost.hip,
# profiles/r03_valu_issue_cost.txt): fp64 FMA / MUL / ADD 2.2 ns, fp64 rcp / rsq / sqrt 6.9 ns, everything else (32-bit moves and
# integer ops ~1.0, fp64 compares / ldexp / div helpers ~1.9) taken at 1.0 ns - a lower bound
VALU_NS = {"fma_mul_add_f64": 2.2, "trans_f64": 6.9, "other": 1.0}


def step_floor(tier, cols, step_ms, many_stream_gbps, fused=False):
    """What the step would take if nothing but the two resources the counters show in use were in the way - VERDICT r03 item 8:
      bytes_ms = sum over the step's kernels of their PMC-measured HBM bytes / the many-stream copy rate of THIS box (measured in
                 this run: 64 + 64 separate 8-byte-per-lane streams, what a many-field kernel sees);
      valu_ms  = sum over the kernels of their VALU wave-instructions x the measured issue cost of their class / 1 024 SIMDs
                 (per-wrapper step's SQ table; the fused step runs the same arithmetic);
      floor    = max(bytes_ms, valu_ms): both resources perfectly overlapped across the whole step;
      serial_floor = sum over kernels of max(bytes_k, valu_k): every kernel at its own roof, no overlap between kernels.
    frac_of_floor = floor / measured: the rest is lack of overlap between the fp64-bound and the bandwidth-bound phases, lanes
    idle in issued instructions, tails and launch gaps.  Only for the build and the column count the tables were measured on."""
    tname = f"{PROFILE_TAG}_hbm_traffic_pmc_{'fused_' if fused else ''}tier{tier}.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    cpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_compute_pmc_tier{tier}.json")
    if not (os.path.exists(tpath) and os.path.exists(cpath)) or not many_stream_gbps:
        return None
    tdoc, cdoc = json.load(open(tpath)), json.load(open(cpath))
    h = kernel_source_hash()
    if tdoc.get("source_hash") != h or cdoc.get("source_hash") != h or int(tdoc.get("columns", 1_000_000)) != cols or cols != 1_000_000:
        return None
    per = {}
    for name, v in tdoc["kernels"].items():
        if name == "elmk::k_copy":
            continue
        per[name.replace("elmk::", "")] = [v["hbm_bytes_per_launch"] / (many_stream_gbps * 1e9) * 1e3, 0.0]
    valu_total = 0.0
    for name, k in cdoc["kernels"].items():
        f64 = k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_ADD_F64"]
        tr = k["SQ_INSTS_VALU_TRANS_F64"]
        ns = f64 * VALU_NS["fma_mul_add_f64"] + tr * VALU_NS["trans_f64"] + max(k["SQ_INSTS_VALU"] - f64 - tr, 0.0) * VALU_NS["other"]
        ms = ns / N_SIMD * 1e-6
        valu_total += ms
        if name in per:
            per[name][1] = ms
    bytes_ms = sum(v[0] for v in per.values())
    serial = sum(max(v) for v in per.values()) if not fused else None
    floor = max(bytes_ms, valu_total)
    return {"bytes_ms": round(bytes_ms, 4), "valu_ms": round(valu_total, 4), "floor_ms": round(floor, 4),
            "serial_floor_ms": None if serial is None else round(serial, 4), "measured_ms": round(step_ms, 4),
            "frac_of_floor": round(floor / step_ms, 4), "many_stream_GBps": many_stream_gbps, "valu_ns_per_wave_instruction": VALU_NS,
            "how": "max(sum of PMC HBM bytes / this box's many-stream copy rate, sum of VALU wave-instructions x measured issue cost / 1024 SIMDs); "
                   "committed tables of this build (profiles/)"}


SOIL_ALGO_BYTES = 2860  # soil_temperature: 1972 B read + 888 B written per column (tally in DESIGN.md section 9)
SOIL_RESTORE = ["t_soisno", "h2osoi_ice", "h2osoi_liq", "t_h2osfc", "h2osfc", "h2osno", "snow_depth", "int_snow", "t_grnd"]


def make_step(D, workload, fused=False):
    if hasattr(D, "rehearsal_step"):  # tests only: the device step is the one thing replaced
        return D.rehearsal_step
    from elmkernels_amd import state as st

    if workload == "timestep7":
        adv = st.timestep7_fused if fused else st.timestep7

        def step():
            D.restore_fields()
            adv(D, 1800.0)
    else:
        def step():
            D.restore_fields()
            st.kokkos_soil_temperature(D, 1800.0)
    return step


PER_RANK_SECONDS = []  # of the last timed_steps call with a process group (rank order)


def timed_steps(D, workload, steps, warmup, sync_all, dist, torch, red_device="cuda", fused=False):
    """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks -> seconds."""
    step = make_step(D, workload, fused)
    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    D.sync()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist is not None:
        # every rank's own seconds (a straggler GPU shows here), then the maximum - the time the job took
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        allt = [torch.zeros(1, dtype=torch.float64, device=red_device) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    PER_RANK_SECONDS[:] = per_rank
    return elapsed


def event_time_soil(D, nsteps):
    """Mean device time of the soil_temperature launch: HIP events on the context's stream (elmk_profile_wrapper), the
    snapshot restored before every launch outside the event brackets."""
    from elmkernels_amd import state as st

    return D.profile_wrapper(st.WRAPPER_NAMES.index("soil_temperature"), 1800.0, nsteps)


def active_bytes(veg_frac, sun_frac):
    """Per-wrapper algorithmic bytes charging a predicate-gated wrapper only for the columns it works on."""
    act = dict(ALGO_BYTES)
    a = GATED_IDLE_BYTES
    act["albedo_snicar"] = a["albedo_snicar"] + sun_frac * (ALGO_BYTES["albedo_snicar"] - a["albedo_snicar"])
    act["bareground_fluxes"] = a["bareground_fluxes"] + (1 - veg_frac) * (ALGO_BYTES["bareground_fluxes"] - a["bareground_fluxes"])
    act["canopy_fluxes"] = a["canopy_fluxes"] + veg_frac * (ALGO_BYTES["canopy_fluxes"] - a["canopy_fluxes"])
    return act


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stop(procs):
    """Terminate the rank processes that are still running (plain signals to our own children; nothing is re-executed)."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.time() + 5.0
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


ALGO_BYTES_FUSED_F32 = 1721  # the fused bound with an fp32 state: 3 368 B of fp64 fields halved + 37 B of int / bool fields


def fp32_state_variant(device_index, seed, st, compact=False):
    """BASELINE config 5, second half: the fused step on an fp32 STATE (libelmk_f32.so: every fp64 field stored as fp32, all
    arithmetic fp64 - widen on load, round on store), 10 M columns (and 1 M), both tiers.  A report: throughput, the roofline
    against the 1 721 B/column-step bound, PMC traffic and the per-kernel register / occupancy table when the committed tables
    belong to this build, and how far the outputs of one step are from the fp64 state's (the same step through libelmk.so on
    the same inputs, on the device).  Never `value`: the fp64 outputs are the contract (1e-12), these are not within it."""
    import numpy as np

    from elmkernels_amd import _lib as L

    if not os.path.exists(L.F32_LIB_PATH):
        return {"error": "libelmk_f32.so not built"}
    out = {"what": "fused step, fp32-stored state (fp64 arithmetic), libelmk_f32.so; report only", "bytes_per_column_step": ALGO_BYTES_FUSED_F32}
    # compact (the default run, so that the driver's command times BASELINE config 5 at its own size): 10 M columns only and a
    # 50 000-column pair for the error percentiles; --state-f32 adds 1 M columns, a 200 000-column pair and the register table
    for cols_n, key in (((NORTH_STAR_COLS, "10M"),) if compact else ((1_000_000, "1M"), (NORTH_STAR_COLS, "10M"))):
        for tier in ("A", "B"):
            D, _ = build_state(cols_n, device_index, tier, seed, lib_path=L.F32_LIB_PATH)
            for _ in range(4 if compact else 6):  # (the canopy scheduling hints settle)
                D.restore_fields()
                st.timestep7_fused(D, 1800.0)
            each = sorted(D.profile_steps(1800.0, 5 if compact else 7, fused=True))
            med = each[len(each) // 2]
            ms, tot = D.profile_timestep7_fused(1800.0, 3 if compact else 5)
            rec = {"value": cols_n / (med * 1e-3), "unit": "gridcell-timesteps/s", "ms_per_step_median_events": med,
                   "frac_of_fp32_fused_bound": ALGO_BYTES_FUSED_F32 * cols_n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "launch_groups_ms": {k: round(m, 4) for k, m in zip(st.KERNEL_NAMES_FUSED, ms)}, "device_state_GB": round(D.device_bytes / 1e9, 3)}
            if cols_n == 1_000_000:
                tr = pmc_traffic_total(f"{PROFILE_TAG}_hbm_traffic_pmc_fused_f32_tier{tier}.json", cols_n)
                rec["traffic_bytes_per_column_step"] = None if tr is None else tr / cols_n
            else:
                tr = pmc_traffic_total(f"{PROFILE_TAG}_hbm_traffic_pmc_10M_fused_f32_tier{tier}.json", cols_n)
                rec["traffic_bytes_per_column_step"] = None if tr is None else tr / cols_n
            D.close()
            out.setdefault(key, {})[TIER_NAMES[tier]] = rec
    # one step from identical inputs through both builds, on the device: how far does the fp32 state move the outputs?
    n = 50_000 if compact else 200_000
    err = {}
    for tier in ("A", "B"):
        D64, _ = build_state(n, device_index, tier, seed)
        D32, _ = build_state(n, device_index, tier, seed, lib_path=L.F32_LIB_PATH)
        st.timestep7_fused(D64, 1800.0)
        st.timestep7_fused(D32, 1800.0)
        rels = []
        worst = {}
        for name in FP32_STUDY_FIELDS:
            a, b = D64[name].astype(np.float64).ravel(), D32[name].astype(np.float64).ravel()
            scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)
            r = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, np.abs(a - b) / scale)
            r = r[np.isfinite(r)]
            rels.append(r)
            worst[name] = float(np.percentile(r, 99)) if r.size else 0.0
        r = np.concatenate(rels)
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
        err[TIER_NAMES[tier]] = {"columns": n, "values": int(r.size), "median": float(np.median(r)), "p90": float(np.percentile(r, 90)),
                                 "p99": float(np.percentile(r, 99)), "max": float(r.max()),
                                 "largest_p99_fields": {k: v for k, v in top}}
        D64.close()
        D32.close()
    out["relative_difference_to_fp64_state_after_one_step"] = err
    occ = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_kernel_resources.json")
    if os.path.exists(occ) and not compact:
        doc = json.load(open(occ))
        if doc.get("source_hash") == kernel_source_hash():
            out["registers_and_occupancy"] = doc["kernels"]
    return out


# outputs of the seven wrappers that the comparison of the two state precisions looks at
FP32_STUDY_FIELDS = ["t_veg", "t_grnd", "h2ocan", "btran", "qflx_tran_veg", "qflx_evap_veg", "qflx_evap_soi", "qflx_evap_tot", "eflx_sh_veg",
                     "eflx_sh_grnd", "eflx_sh_tot", "eflx_lh_tot", "cgrnd", "cgrnds", "cgrndl", "t_ref2m", "q_ref2m", "rh_ref2m", "dlrad",
                     "ulrad", "albd", "albi", "fabd", "fabi", "sabg", "sabv", "fsa", "fsr", "frac_sno", "snow_depth", "h2osno", "qg", "thm",
                     "tssbef", "rootr", "eff_porosity", "fwet", "fdry"]


def pmc_traffic_total(table, cols=None):
    """Sum over all physics kernels of a committed PMC traffic table (bytes per launch), if it belongs to this build (and, when
    cols is given, to that column count)."""
    path = os.path.join(ROOT, "profiles", table)
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if doc.get("source_hash") not in (None, kernel_source_hash()):
        return None
    if cols is not None and int(doc.get("columns", 1_000_000)) != cols:
        return None
    return float(sum(v["hbm_bytes_per_launch"] for k, v in doc["kernels"].items() if k != "elmk::k_copy"))


def spawn_ranks(ngpus, argv, timeout_s=None, poll_s=0.2):
    """`python bench.py --gpus N` with no launcher: start N rank processes (one per GPU), relay rank 0's JSON line.
    The parent initialises nothing on the GPU (torch is not even imported here) and never re-executes itself.  It polls its
    children: the first one that exits non-zero (or an overall timeout, ELMK_BENCH_TIMEOUT seconds, default 3000) ends the
    others within seconds and becomes the exit code - a rank that dies before the barrier must not leave the rest waiting
    in the rendezvous for the backend's timeout.  The rendezvous port is picked by bind-and-close; if another process takes
    it before rank 0 listens, the ranks fail at once with EADDRINUSE and the spawn is retried on a fresh port."""
    import tempfile

    timeout_s = float(os.environ.get("ELMK_BENCH_TIMEOUT", "3000")) if timeout_s is None else timeout_s
    for attempt in range(3):
        port = _free_port()
        out0 = tempfile.TemporaryFile()
        err0 = tempfile.TemporaryFile()
        procs = []
        for r in range(ngpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err0 if r == 0 else None))
        t0 = time.time()
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = abs(bad[0][1]) or 1
                print(f"bench.py: rank {bad[0][0]} exited with {bad[0][1]}: stopping the other ranks", file=sys.stderr)
                _stop(procs)
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > timeout_s:
                print(f"bench.py: ranks still running after {timeout_s:.0f} s: stopping them", file=sys.stderr)
                _stop(procs)
                rc = 124
                break
            time.sleep(poll_s)
        err0.seek(0)
        err_text = err0.read().decode(errors="replace")
        if rc and attempt < 2 and time.time() - t0 < 60 and ("EADDRINUSE" in err_text or "address already in use" in err_text.lower()):
            print("bench.py: rendezvous port was taken: retrying on another port", file=sys.stderr)
            continue
        sys.stderr.write(err_text)
        out0.seek(0)
        for line in out0.read().decode(errors="replace").splitlines():
            # stdout carries the ONE JSON line; anything else a library printed there (gloo's connection banner) goes to stderr
            (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
        sys.stdout.flush()
        return rc
    return 1


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cols", type=int, default=1_000_000,
                    help="columns per GPU (BASELINE config 2: 1M; config 4 is --gpus 8 --cols 10000000)")
    ap.add_argument("--tier", default="A", choices=["A", "B"],
                    help="synthetic state (SURVEY.md 8(d)): A fixture-tiled (default), B branch-mix (snow layers, bare ground, C4 ...)")
    ap.add_argument("--workload", default="timestep7", choices=["timestep7", "soil_temperature"],
                    help="timestep7: BASELINE config 2 (the 7 wrappers); soil_temperature: config 3 (the soil-column vertical solve)")
    ap.add_argument("--fused", action="store_true",
                    help="the step through elmk_timestep7_fused (BASELINE config 5's launch structure, fp64 state): "
                         "roofline against the 3 405 B/column-step fused bound")
    ap.add_argument("--no-other-tier", action="store_true", help="skip the secondary measurement on the other tier")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 10 M-column measurement (north_star_10M)")
    ap.add_argument("--seed", type=int, default=0x5EEDE1A0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the seven wrappers as one captured HIP graph (elmk_set_graph): removes host launch latency, "
                         "which dominates below ~100k columns")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the per-kernel HIP-event profile")
    ap.add_argument("--one-gpu-value", type=float, default=None,
                    help="the 1-GPU `value` of the same configuration: with it the line carries scaling_efficiency = value / (N * that)")
    ap.add_argument("--no-soil-10m", action="store_true", help="skip the soil-column solve at 10 M columns (soil_temperature_10M)")
    ap.add_argument("--no-two-blocks", action="store_true", help="skip the two-block pipeline measurement (two_block_pipeline)")
    ap.add_argument("--no-state-f32", action="store_true", help="skip the compact fp32-state measurement of the default run (fp32_state_10M)")
    ap.add_argument("--state-f32", action="store_true",
                    help="also measure BASELINE config 5's fp32-state variant (libelmk_f32.so: fp64 fields stored as fp32, fp64 arithmetic, "
                         "fused step) as a separate object fp32_state_10M - reported beside the fp64 numbers, never as `value`")
    args = ap.parse_args(argv)

    if "RANK" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world  # under a launcher the launcher decides

    # every HIP stream its own hardware queue (the runtime's default is 4 queues shared by all streams of the process): kernels of two
    # contexts' streams can then be resident together (two_block_pipeline); no effect on the one-context measurements (A/B in
    # profiles/r04_two_block_overlap.txt).  Read by the runtime when it initialises, hence before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if world == 1 and not args.no_cpu_baseline:
        # the CPU-baseline leg: OpenMP threads bound to neighbouring cores and sleeping between parallel regions instead of
        # spinning on a shared host.  libgomp reads these once, when it is loaded (import torch loads it), hence here.  Not at
        # N > 1: binding would pin every rank's launching thread to the first core of the same mask.
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    import torch

    rehearsal = None
    if os.environ.get(REHEARSAL_ENV):
        import importlib

        mod, fn = os.environ[REHEARSAL_ENV].split(":")
        rehearsal = getattr(importlib.import_module(mod), fn)

    dist = None
    ndev = torch.cuda.device_count()
    if ndev <= 0 and rehearsal is None:
        sys.exit("bench.py needs a HIP device (there is no CPU path)")
    # one process per GPU; a rehearsal with more ranks than GPUs (several ranks share a card) cannot use RCCL and
    # falls back to gloo for the barrier / max-reduce, which is all this benchmark communicates
    shared = world > ndev
    device_index = local_rank % ndev if ndev > 0 else 0
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        if ndev > 0:
            torch.cuda.set_device(device_index)
        if shared:
            # gloo announces its connections on stdout (C++ side): keep stdout for the one JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    red_device = "cpu" if shared else "cuda"

    from elmkernels_amd import decomp

    st = None
    if rehearsal is None:
        from elmkernels_amd import state as st

    # weak scaling: every rank owns args.cols columns; the global problem is the 1-D block split of world*cols
    ncols_global = args.cols * world
    start, ncols = decomp.block_range(ncols_global, world, rank)
    soil = args.workload == "soil_temperature"
    make = rehearsal or build_state

    def prepared(tier, n=None):
        D, host_state = make(ncols if n is None else n, device_index, tier, args.seed + rank)
        if soil and rehearsal is None:  # the solve follows the seven wrappers: run them once, then keep the state the solve starts from
            st.timestep7(D, 1800.0)
            D.snapshot_fields(SOIL_RESTORE)
            D.sync()
        return D, host_state

    def measure(D, steps, warmup, with_dist, fused=None):
        """-> (seconds of the timed region, per-launch-group ms, ms of the whole step by HIP events)."""
        fused = args.fused if fused is None else fused
        def sync_all():
            D.sync()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            if with_dist and dist is not None:
                dist.barrier()

        el = timed_steps(D, args.workload, steps, warmup, sync_all, dist if with_dist else None, torch, red_device, fused)
        if rehearsal is not None:
            return el, [], 0.0
        if soil:
            return el, [], event_time_soil(D, max(1, args.profile_steps))
        D.restore_fields()
        ms, tot = (D.profile_timestep7_fused if fused else D.profile_timestep7)(1800.0, max(1, args.profile_steps))
        return el, ms, tot

    def fused_too(D, n, steps, warmup):
        """The same state through elmk_timestep7_fused (reported beside the per-wrapper step, never instead of it)."""
        el, msf, totf = measure(D, steps, warmup, False, fused=True)
        return {"value": n * steps / el, "ms_per_step": el / steps * 1e3, "ms_per_step_events": totf,
                "frac_of_fused_bound": ALGO_BYTES_FUSED * n / (totf * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "launch_groups_ms": {k: round(m, 4) for k, m in zip(st.KERNEL_NAMES_FUSED, msf)}}

    def advance_too(D, n):
        """The whole device part of ELMInterface::advance as one call (elmk_advance_physics: the fused seven, soil_temperature,
        snow_hydrology, surface_fluxes): mean device time per step by HIP events, with the same restores between steps as the
        seven-wrapper step.  Reported beside that step - BASELINE's metric is defined on the seven - never instead of it."""
        from elmkernels_amd import synth

        D.set_snow_age_tables(synth.snow_age_tables())
        D.profile_wrapper(st.WRAPPER_NAMES.index("advance_physics"), 1800.0, 5)  # (warm-up: scheduling hints, first-step transients)
        ms = D.profile_wrapper(st.WRAPPER_NAMES.index("advance_physics"), 1800.0, 5)
        return {"ms_per_step_events": round(ms, 4), "value": n / (ms * 1e-3)}

    D, host_state = prepared(args.tier)
    # population of the predicate-gated wrappers ("active-bytes" variant of the roofline numerator)
    veg_frac = float((D["frac_veg_nosno"] != 0).mean())
    sun_frac = float((D["coszen"] > 0).mean())
    if args.graph and not soil and rehearsal is None:
        D.set_graph(True)
    elapsed, ms, ms_total = measure(D, args.steps, args.warmup, True)
    flags, first_bad = D.error_summary()
    state_gb = round(D.device_bytes / 1e9, 3)
    step_time = None
    empirical = None
    if rehearsal is None and rank == 0:
        if not soil:
            # device time of every step (HIP events around each, same restores): SURVEY 8(d) asks for the median
            each = sorted(D.profile_steps(1800.0, max(3, args.steps), fused=args.fused))
            step_time = {"n": len(each), "median_ms": each[len(each) // 2] if len(each) % 2 else 0.5 * (each[len(each) // 2 - 1] + each[len(each) // 2]),
                         "mean_ms": sum(each) / len(each), "min_ms": each[0], "max_ms": each[-1], "how": "HIP events around each step"}
        if world == 1:
            # the empirical HBM line of this box beside the datasheet peak: a device-to-device copy of 1 GiB in four access
            # shapes (the best is the line), and the same bytes as 64 + 64 separate streams (what a many-field kernel sees)
            shapes = {sh: D.copy_bandwidth(1 << 30, 10, sh) for sh in range(5)}
            empirical = {"copy_GBps_by_shape": {"8B_per_lane": round(shapes[0], 1), "16B_per_lane": round(shapes[1], 1),
                                                "8B_per_lane_x4_loads": round(shapes[2], 1), "16B_per_lane_x4_loads": round(shapes[3], 1)},
                         "peak_GBps": round(max(shapes[k] for k in range(4)), 1), "many_stream_GBps": round(shapes[4], 1)}
    names = [] if rehearsal is not None else (st.KERNEL_NAMES_FUSED if args.fused else st.KERNEL_NAMES)
    also_fused = rank == 0 and world == 1 and rehearsal is None and not args.fused and not (args.workload == "soil_temperature")
    fused_step = None
    advance_step = None
    if also_fused:
        fused_step = {"bytes_per_column_step": ALGO_BYTES_FUSED, TIER_NAMES[args.tier]: fused_too(D, ncols, args.steps, args.warmup)}
        if args.cols <= 2_000_000:
            advance_step = {"what": "elmk_advance_physics: the seven (fused) + soil_temperature + snow_hydrology + surface_fluxes, one call",
                            TIER_NAMES[args.tier]: advance_too(D, ncols)}

    other = None
    north = None
    solo = rank == 0 and world == 1 and rehearsal is None
    if solo and not args.no_other_tier:
        D.close()
        D = None
        ot = "B" if args.tier == "A" else "A"
        D2, _ = prepared(ot)
        if args.graph and not soil:
            D2.set_graph(True)
        el2, _, tot2 = measure(D2, args.steps, args.warmup, False)
        other = {"tier": TIER_NAMES[ot], "value": ncols_global * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3,
                 "ms_per_step_events": tot2}
        if also_fused:
            fused_step[TIER_NAMES[ot]] = fused_too(D2, ncols, args.steps, args.warmup)
            if advance_step is not None:
                advance_step[TIER_NAMES[ot]] = advance_too(D2, ncols)
        D2.close()
    if solo and not args.no_north_star and not soil and args.cols < NORTH_STAR_COLS:
        # the north-star size in the same run: 10 M columns (64 GB of state + scratch), 5 timed steps per tier
        if D is not None:
            D.close()
            D = None
        north = {"columns": NORTH_STAR_COLS, "steps": 5, "warmup": 2}
        byts = ALGO_BYTES_FUSED if args.fused else ALGO_BYTES_STEP
        for tier in ("A", "B"):
            Dn, _ = prepared(tier, NORTH_STAR_COLS)
            vf = float((Dn["frac_veg_nosno"] != 0).mean())
            sf = float((Dn["coszen"] > 0).mean())
            eln, msn, totn = measure(Dn, 5, 2, False)
            act = sum(active_bytes(vf, sf).values())
            north[TIER_NAMES[tier]] = {
                "value": NORTH_STAR_COLS * 5 / eln, "ms_per_step": eln / 5 * 1e3, "ms_per_step_events": totn,
                "timestep_roofline": {
                    "bytes_per_column_step": byts, "frac": byts * NORTH_STAR_COLS / (totn * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_active_bytes": None if args.fused else act * NORTH_STAR_COLS / (totn * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "kernels_ms": {n: round(m, 4) for n, m in zip(names, msn)},
                "device_state_GB": round(Dn.device_bytes / 1e9, 3),
            }
            # PMC HBM bytes of the whole step AT THIS SIZE (profiles/<round>_hbm_traffic_pmc_10M_*: rocprofv3 --pmc passes at 10 M
            # columns, reported only for the build they were measured on)
            tr = pmc_traffic_total(f"{PROFILE_TAG}_hbm_traffic_pmc_10M_{'fused_' if args.fused else ''}tier{tier}.json", NORTH_STAR_COLS)
            north[TIER_NAMES[tier]]["timestep_roofline"]["traffic_bytes_per_column_step"] = None if tr is None else round(tr / NORTH_STAR_COLS, 1)
            if also_fused:
                north[TIER_NAMES[tier]]["fused_step"] = fused_too(Dn, NORTH_STAR_COLS, 5, 2)
                tr = pmc_traffic_total(f"{PROFILE_TAG}_hbm_traffic_pmc_10M_fused_tier{tier}.json", NORTH_STAR_COLS)
                north[TIER_NAMES[tier]]["fused_step"]["traffic_bytes_per_column_step"] = None if tr is None else round(tr / NORTH_STAR_COLS, 1)
            Dn.close()

    soil10 = None
    if solo and not soil and not args.no_soil_10m and not args.no_north_star and args.cols < NORTH_STAR_COLS:
        # BASELINE config 3 in the same run: the soil-column vertical solve at 10 M columns (branch-mix tier: resolved snow
        # layers, ponded water, frozen soil), state left by the seven wrappers, 5 solves timed by HIP events
        if D is not None:
            D.close()
            D = None
        Ds, _ = build_state(NORTH_STAR_COLS, device_index, "B", args.seed)
        st.timestep7(Ds, 1800.0)
        Ds.snapshot_fields(SOIL_RESTORE)
        Ds.sync()
        event_time_soil(Ds, 2)
        ms10 = event_time_soil(Ds, 5)
        gbs10 = SOIL_ALGO_BYTES * NORTH_STAR_COLS / (ms10 * 1e-3) / 1e9
        soil10 = {"workload": "soil-column vertical solve (kokkos_soil_temperature), 10 M columns, branch-mix tier, fp64", "columns": NORTH_STAR_COLS,
                  "value": NORTH_STAR_COLS / (ms10 * 1e-3), "unit": "columns/s", "ms_per_solve_events": ms10,
                  "roofline": {"bound": "hbm", "kernel": "k_soil_temperature", "achieved": gbs10, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs10 / HBM_PEAK_GBS, "bytes_per_column": SOIL_ALGO_BYTES}}
        Ds.close()

    blocks2 = None
    if solo and not soil and not args.no_two_blocks and args.cols >= 524288 and args.cols <= 4_000_000:
        # The same columns as TWO blocks - two contexts of cols / 2 on their own streams, the leaf-temperature iteration in
        # 256-thread workgroups (ELMK_OPT_CF_HALF_WORKGROUPS) so that one block's streaming kernels are resident on the CUs beside the
        # other block's fp64-bound iteration; fused steps enqueued back to back like the main measurement.  What overlapping the two
        # halves of the step buys; reported beside `fused_step`, never as `value`.
        if D is not None:
            D.close()
            D = None
        blocks2 = {"what": "fused step, the columns as two contexts of cols / 2 on two streams, k_cf_iterate in 256-thread workgroups "
                           "(elmk_set_option ELMK_OPT_CF_HALF_WORKGROUPS); needs GPU_MAX_HW_QUEUES above the runtime's default of 4",
                   "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
        for tier in ("A", "B"):
            half = args.cols // 2
            ctxs = [build_state(half, device_index, tier, args.seed + i)[0] for i in range(2)]
            for Dk in ctxs:
                Dk.set_option(st.OPT_CF_HALF_WORKGROUPS, 1)

            def both():
                for Dk in ctxs:
                    Dk.restore_fields()
                    st.timestep7_fused(Dk, 1800.0)

            for _ in range(max(args.warmup, 6)):
                both()
            for Dk in ctxs:
                Dk.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                both()
            for Dk in ctxs:
                Dk.sync()
            el = time.perf_counter() - t0
            for Dk in ctxs:
                Dk.close()
            ref = fused_step.get(TIER_NAMES[tier], {}).get("value") if fused_step else None
            blocks2[TIER_NAMES[tier]] = {"value": 2 * half * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                                         "vs_one_context_fused": None if not ref else 2 * half * args.steps / el / ref}

    f32 = None
    if solo and not soil and (args.state_f32 or (not args.no_north_star and not args.no_state_f32 and args.cols < NORTH_STAR_COLS)):
        if D is not None:
            D.close()
            D = None
        f32 = fp32_state_variant(device_index, args.seed, st, compact=not args.state_f32)
        if "error" in f32 and not args.state_f32:
            f32 = None  # (libelmk_f32.so not built: the default run simply has no config-5 object)

    if rank == 0:
        value = ncols_global * args.steps / elapsed
        tier_name = TIER_NAMES[args.tier]
        out = {
            "metric": "gridcell-timesteps/sec",
            "value": value,
            "unit": "gridcell-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
        }
        if world > 1:
            # every rank's own time for the K steps (ms per step): a straggler GPU is visible here, the job's time is the maximum
            out["per_rank_ms"] = [round(t / args.steps * 1e3, 4) for t in PER_RANK_SECONDS]
            if args.one_gpu_value:
                out["scaling_efficiency"] = value / (world * args.one_gpu_value)
        if step_time is not None:
            out["step_time"] = step_time
        par = f"columns block-split over {world} rank(s) on {min(world, max(ndev, 1))} GPU(s), no collective"
        if rehearsal is not None:
            out["config"] = {"workload": "CPU rehearsal of the multi-rank control flow (tests only)", "columns_per_gpu": args.cols,
                             "columns_total": ncols_global, "parallelism": par}
        elif soil:
            gbs = SOIL_ALGO_BYTES * ncols / (ms_total * 1e-3) / 1e9
            out["config"] = {
                "workload": f"soil-column vertical solve (kokkos_soil_temperature: 21-row pentadiagonal system, phase change), {args.cols} columns per GPU, fp64",
                "columns_per_gpu": args.cols, "columns_total": ncols_global, "levels": 20, "tier": tier_name, "parallelism": par,
            }
            out["roofline"] = {"bound": "hbm", "kernel": "k_soil_temperature", "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "traffic": pmc_traffic("soil_temperature", args.tier, args.cols, f"{PROFILE_TAG}_hbm_traffic_pmc_soil_tier{args.tier}.json"),
                               "bytes_per_column": SOIL_ALGO_BYTES, "avg_launch_ms": ms_total}
            if empirical is not None:
                out["roofline"]["empirical_peak"] = empirical["peak_GBps"]
                out["roofline"]["frac_of_empirical_peak"] = gbs / empirical["peak_GBps"]
                out["roofline"]["empirical"] = empirical
        else:
            act = active_bytes(veg_frac, sun_frac)
            kern = {}
            for name, m in zip(names, ms):
                kern[name] = {"ms": round(m, 4)}
                if name in ALGO_BYTES:
                    # charged only for the columns the wrapper works on (a gated wrapper with nothing to do would
                    # otherwise show more than the peak)
                    gbs = act[name] * ncols / (m * 1e-3) / 1e9 if m > 0 else 0.0
                    kern[name].update({"active_bytes_per_column": round(act[name], 1), "algo_GBps": round(gbs, 1),
                                       "frac_of_peak": round(min(gbs / HBM_PEAK_GBS, 1.0), 4)})
            dom = max(zip(names, ms), key=lambda x: x[1])
            if args.fused:
                step_bytes = ALGO_BYTES_FUSED
                # the fused step's launch groups: frac_wet + albedo (960 + 44 B), the streaming stage (what is left of the
                # 3 405 B bound), the bare-ground list and the leaf-temperature iteration + finish (charged to the stream)
                dom_bytes = {"albedo_snicar": ALGO_BYTES["albedo_snicar"], "fused_stream": ALGO_BYTES_FUSED - ALGO_BYTES["albedo_snicar"],
                             "canopy_iterate": ALGO_BYTES["canopy_fluxes"]}.get(dom[0], ALGO_BYTES.get(dom[0], 0))
            else:
                step_bytes = ALGO_BYTES_STEP
                dom_bytes = ALGO_BYTES[dom[0]]
            dom_gbs = dom_bytes * ncols / (dom[1] * 1e-3) / 1e9
            step_gbs = step_bytes * ncols / (ms_total * 1e-3) / 1e9
            out["config"] = {
                "workload": f"full water+energy timestep (7 kernels{', fused streaming stage' if args.fused else ''}), {args.cols} columns x 20 soil+snow levels per GPU, fp64",
                "columns_per_gpu": args.cols, "columns_total": ncols_global, "levels": 20, "tier": tier_name, "parallelism": par,
            }
            # which roof applies, from evidence: the committed SQ counter table of this build says how busy the vector ALUs
            # are in the kernels that are not streaming kernels (canopy_fluxes' leaf-temperature iteration, SNICAR)
            comp = compute_roofline(["k_cf_iterate", "k_alb_snicar<1>", "k_bg_flux", "k_cf_init", "k_alb_final"], args.tier, args.cols)
            dom_compute = {"canopy_fluxes": "k_cf_iterate", "canopy_iterate": "k_cf_iterate", "albedo_snicar": "k_alb_snicar<1>"}.get(dom[0])
            # `bound` names the roof achieved / peak / frac are measured against (always the HBM line here: the path has no
            # contraction); `limited_by` says what the counters show the dominant kernel actually waiting for
            limited_by = "hbm"
            if comp and dom_compute in comp and comp[dom_compute]["valu_busy"] > 0.70:
                limited_by = "fp64_valu"
            out["roofline"] = {
                "bound": "hbm", "limited_by": limited_by, "kernel": f"{dom[0]} ({', '.join(p.replace('elmk::', '') + '*' for p in KERNEL_PREFIX.get(dom[0], ()))})",
                "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom[0], args.tier, args.cols,
                                       f"{PROFILE_TAG}_hbm_traffic_pmc_fused_tier{args.tier}.json" if args.fused else None),
                "bytes_per_column": dom_bytes, "avg_launch_ms": dom[1],
                "bound_evidence": (f"{dom_compute}: VALU busy {comp[dom_compute]['valu_busy']:.2f} of the cycles (committed SQ counter pass of this build)"
                                   if comp and dom_compute in comp else "no counter table for this build: bound not established"),
            }
            if empirical is not None:
                out["roofline"]["empirical_peak"] = empirical["peak_GBps"]
                out["roofline"]["frac_of_empirical_peak"] = dom_gbs / empirical["peak_GBps"]
                out["roofline"]["empirical"] = empirical
                fl = step_floor(args.tier, args.cols, ms_total, empirical["many_stream_GBps"], fused=args.fused)
                if fl is not None:
                    out["roofline"]["floor"] = fl
            if comp and limited_by == "fp64_valu":
                c = comp[dom_compute]
                out["roofline"]["compute"] = {"kernel": dom_compute, "achieved": c["fp64_tflops"], "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                              "frac": c["frac_of_fp64_peak"], "valu_busy": c["valu_busy"], "valu_lane_util": c["valu_lane_util"]}
            if comp:
                out["compute_roofline"] = {"peak_fp64_vector_TFLOPs": FP64_VECTOR_PEAK_TFLOPS, "kernels": comp}
            out["timestep_roofline"] = {
                "bytes_per_column_step": step_bytes, "achieved_GBps": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                "ms_per_step_events": ms_total,
                # the unfused tally charging a gated wrapper only for the columns it works on
                "active_bytes_per_column_step": round(sum(act.values()), 1),
                "frac_active_bytes": None if args.fused else sum(act.values()) * ncols / (ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "sunlit_fraction": round(sun_frac, 4), "vegetated_fraction": round(veg_frac, 4),
            }
            out["kernels"] = kern
        out["error_flags"] = flags
        out["device_state_GB"] = state_gb
        if other is not None:
            out["other_tier"] = other
        if fused_step is not None:
            out["fused_step"] = fused_step
        if advance_step is not None:
            out["advance_step"] = advance_step
        if north is not None:
            out["north_star_10M"] = north
        if soil10 is not None:
            out["soil_temperature_10M"] = soil10
        if blocks2 is not None:
            out["two_block_pipeline"] = blocks2
        if f32 is not None:
            out["fp32_state_10M"] = f32
        if not args.no_cpu_baseline and world == 1 and rehearsal is None:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(host_state, workload=args.workload)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if D is not None:
        D.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
