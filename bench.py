#!/usr/bin/env python3
"""bench.py - gridcell-timesteps/sec of the full water+energy timestep (all 7 kernels) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cols C] [--tier A|B] [--workload timestep7|soil_temperature]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Columns shard halo-free across ranks (elmkernels_amd.decomp: the reference's 1-D block
split); there is no data-path collective - torch.distributed (RCCL) only carries the barrier and the
max-over-ranks of the timed region.  Per-GPU work is fixed (weak scaling): --cols columns on every rank.

A "step" = one pass of the reference's ELMInterface::advance hot path over the resident state: restore of the
snapshot fields (t_veg + forcing heights: what the rest of the model does between steps, 64 B/column), then
frac_wet -> albedo_snicar -> canopy_hydrology -> surface_radiation -> canopy_temperature -> bareground_fluxes
-> canopy_fluxes, each a hand-written HIP kernel over SoA state in HBM.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timed on the launch stream) and `cpu_baseline` (the C oracle with OpenMP on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per column-step (SURVEY.md section 8(d) / BASELINE.md section 3; DESIGN.md restates the tally)
ALGO_BYTES = {
    "frac_wet": 44, "albedo_snicar": 960, "canopy_hydrology": 440, "surface_radiation": 640,
    "canopy_temperature": 753, "bareground_fluxes": 368, "canopy_fluxes": 2076,
}
ALGO_BYTES_STEP = sum(ALGO_BYTES.values())  # 5281
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
RESTORE_FIELDS = ["t_veg", "forc_hgt_u_patch", "forc_hgt_t_patch", "forc_hgt_q_patch"]
NBASE = 47 * 64 * 8  # host-generated base block; the device tiles it to --cols


def build_state(ncols, device, tier, seed):
    from elmkernels_amd import state as st
    from elmkernels_amd import synth

    ft = st.field_table()
    nbase = min(NBASE, ncols)
    cols, scal, soil = synth.make_state(ft, nbase, tier=tier, seed=seed)
    D = st.ELMState(ncols, device)
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


def cpu_baseline(host_state, budget_s=15.0, workload="timestep7"):
    """The oracle (plain-C restatement of the reference physics, OpenMP over columns like Kokkos-OpenMP) on the
    host cores, on a bounded sample of the same workload."""
    from tests import helpers as H

    cols, scal, soil = host_state
    from oracle import oracle as O

    threads = O.lib().lib.elmo_get_max_threads()
    nb = next(iter(cols.values())).shape[0]
    n = min(nb, 24064)
    sub = {k: v[:n] for k, v in cols.items()}
    S = H.oracle_state(sub, scal, soil)
    tveg = S["t_veg"].copy()
    hg = {k: S[k].copy() for k in RESTORE_FIELDS[1:]}
    S.timestep7(1800.0)  # warm-up (thread pool, page faults)
    if workload == "soil_temperature":
        saved = {k: S[k].copy() for k in SOIL_RESTORE}
    steps = 0
    t0 = time.perf_counter()
    while True:
        if workload == "soil_temperature":
            for k, v in saved.items():
                S[k][...] = v
            S.soil_temperature(1800.0)
        else:
            S["t_veg"][:] = tveg
            for k, v in hg.items():
                S[k][:] = v
            S.timestep7(1800.0)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 5000:
            break
    return {
        "value": n * steps / el, "unit": "gridcell-timesteps/s", "cores": int(threads), "kind": "port",
        "sample": f"{n} columns x {steps} timesteps of the same tier, oracle/libelmoracle.so (gcc -O2 -fopenmp), {el:.1f} s",
    }


# HIP kernels behind each wrapper (names as rocprofv3 reports them)
SUB_KERNELS = {
    "frac_wet": ["elmk::k_frac_wet"],
    "albedo_snicar": ["elmk::k_alb_main"] + [f"elmk::k_alb_snow<{i}>" for i in range(6)],
    "canopy_hydrology": ["elmk::k_canopy_hydrology"],
    "surface_radiation": ["elmk::k_surface_radiation"],
    "canopy_temperature": ["elmk::k_canopy_temperature"],
    "bareground_fluxes": ["elmk::k_bg_main", "elmk::k_bg_flux"],
    "canopy_fluxes": ["elmk::k_cf_count", "elmk::k_cf_init", "elmk::k_cf_iterate", "elmk::k_cf_finish"],
}


def pmc_traffic(wrapper, args):
    """HBM bytes per launch of the wrapper's kernels from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    calibrated on gfx950, WRITE_SIZE as is; profiles/r01_hbm_traffic_pmc_tier{A,B}.json).  The counters cannot be read
    from inside this process, so the number is only reported for the configuration it was measured on."""
    path = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_pmc_tier{args.tier}.json")
    if args.cols != 1_000_000 or not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"]
    try:
        return float(sum(k[name]["hbm_bytes_per_launch"] for name in SUB_KERNELS[wrapper]))
    except KeyError:
        return None


SOIL_ALGO_BYTES = 2860  # soil_temperature: 1972 B read + 888 B written per column (tally in DESIGN.md section 9)
SOIL_RESTORE = ["t_soisno", "h2osoi_ice", "h2osoi_liq", "t_h2osfc", "h2osfc", "h2osno", "snow_depth", "int_snow", "t_grnd"]


def timed_steps(D, workload, steps, warmup, sync_all, dist, torch, red_device="cuda"):
    """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks -> seconds."""
    from elmkernels_amd import state as st

    if workload == "timestep7":
        def step():
            D.restore_fields()
            st.timestep7(D, 1800.0)
    else:
        def step():
            D.restore_fields()
            st.kokkos_soil_temperature(D, 1800.0)
    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    D.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def event_time_soil(D, nsteps):
    """Mean device time of the soil_temperature launch: HIP events on the context's stream (elmk_profile_wrapper), the
    snapshot restored before every launch outside the event brackets."""
    from elmkernels_amd import state as st

    return D.profile_wrapper(st.WRAPPER_NAMES.index("soil_temperature"), 1800.0, nsteps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cols", type=int, default=1_000_000, help="columns per GPU (BASELINE config 2: 1M)")
    ap.add_argument("--tier", default="A", choices=["A", "B"],
                    help="synthetic state (SURVEY.md 8(d)): A fixture-tiled (default), B branch-mix (snow layers, bare ground, C4 ...)")
    ap.add_argument("--workload", default="timestep7", choices=["timestep7", "soil_temperature"],
                    help="timestep7: BASELINE config 2 (the 7 wrappers); soil_temperature: config 3 (the soil-column vertical solve)")
    ap.add_argument("--no-other-tier", action="store_true", help="skip the secondary measurement on the other tier")
    ap.add_argument("--seed", type=int, default=0x5EEDE1A0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the seven wrappers as one captured HIP graph (elmk_set_graph): removes host launch latency, "
                         "which dominates below ~100k columns")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the per-kernel HIP-event profile")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch

    dist = None
    ndev = torch.cuda.device_count()
    if ndev <= 0:
        sys.exit("bench.py needs a HIP device (there is no CPU path)")
    # one process per GPU; a rehearsal with more ranks than GPUs (several ranks share a card) cannot use RCCL and
    # falls back to gloo for the barrier / max-reduce, which is all this benchmark communicates
    shared = world > ndev
    device_index = local_rank % ndev
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        torch.cuda.set_device(device_index)
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    red_device = "cpu" if shared else "cuda"

    from elmkernels_amd import decomp
    from elmkernels_amd import state as st

    # weak scaling: every rank owns args.cols columns; the global problem is the 1-D block split of world*cols
    ncols_global = args.cols * world
    start, ncols = decomp.block_range(ncols_global, world, rank)
    soil = args.workload == "soil_temperature"

    def prepared(tier):
        D, host_state = build_state(ncols, device_index, tier, args.seed + rank)
        if soil:  # the solve follows the seven wrappers: run them once, then keep the state the solve starts from
            st.timestep7(D, 1800.0)
            D.snapshot_fields(SOIL_RESTORE)
            D.sync()
        return D, host_state

    D, host_state = prepared(args.tier)
    # population of the predicate-gated wrappers ("active-bytes" variant of the roofline numerator)
    veg_frac = float((D["frac_veg_nosno"] != 0).mean())
    sun_frac = float((D["coszen"] > 0).mean())
    if args.graph and not soil:
        D.set_graph(True)

    def sync_all():
        D.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    elapsed = timed_steps(D, args.workload, args.steps, args.warmup, sync_all, dist, torch, red_device)

    # per-kernel device time, HIP events recorded on the stream the kernels are launched on
    if soil:
        ms, ms_total = [], event_time_soil(D, max(1, args.profile_steps))
    else:
        D.restore_fields()
        ms, ms_total = D.profile_timestep7(1800.0, max(1, args.profile_steps))
    flags, first_bad = D.error_summary()
    state_gb = round(D.device_bytes / 1e9, 3)

    other = None
    if rank == 0 and world == 1 and not args.no_other_tier:
        D.close()
        D = None
        ot = "B" if args.tier == "A" else "A"
        D2, _ = prepared(ot)
        if args.graph and not soil:
            D2.set_graph(True)

        def sync2():
            D2.sync()
            torch.cuda.synchronize()

        el2 = timed_steps(D2, args.workload, args.steps, args.warmup, sync2, None, torch)
        other = {"tier": {"A": "fixture-tiled", "B": "branch-mix"}[ot], "value": ncols_global * args.steps / el2,
                 "ms_per_step": el2 / args.steps * 1e3}
        D2.close()

    if rank == 0:
        value = ncols_global * args.steps / elapsed
        tier_name = {"A": "fixture-tiled", "B": "branch-mix"}[args.tier]
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
        if soil:
            gbs = SOIL_ALGO_BYTES * ncols / (ms_total * 1e-3) / 1e9
            out["config"] = {
                "workload": f"soil-column vertical solve (kokkos_soil_temperature: 21-row pentadiagonal system, phase change), {args.cols} columns per GPU, fp64",
                "columns_per_gpu": args.cols, "columns_total": ncols_global, "levels": 20, "tier": tier_name,
                "parallelism": f"columns block-split over {world} rank(s) on {min(world, ndev)} GPU(s), no collective",
            }
            out["roofline"] = {"bound": "hbm", "kernel": "k_st_props + k_soil_temperature", "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                               "bytes_per_column": SOIL_ALGO_BYTES, "avg_launch_ms": ms_total}
        else:
            kern = {}
            for name, m in zip(st.KERNEL_NAMES, ms):
                gbs = ALGO_BYTES[name] * ncols / (m * 1e-3) / 1e9 if m > 0 else 0.0
                kern[name] = {"ms": round(m, 4), "algo_GBps": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
            dom = max(zip(st.KERNEL_NAMES, ms), key=lambda x: x[1])
            dom_gbs = ALGO_BYTES[dom[0]] * ncols / (dom[1] * 1e-3) / 1e9
            step_gbs = ALGO_BYTES_STEP * ncols / (ms_total * 1e-3) / 1e9
            out["config"] = {
                "workload": f"full water+energy timestep (7 kernels), {args.cols} columns x 20 soil+snow levels per GPU, fp64",
                "columns_per_gpu": args.cols, "columns_total": ncols_global, "levels": 20, "tier": tier_name,
                "parallelism": f"columns block-split over {world} rank(s) on {min(world, ndev)} GPU(s), no collective",
            }
            out["roofline"] = {
                "bound": "hbm",
                "kernel": " + ".join(n.replace("elmk::", "") for n in SUB_KERNELS[dom[0]]),
                "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom[0], args), "bytes_per_column": ALGO_BYTES[dom[0]], "avg_launch_ms": dom[1],
            }
            out["timestep_roofline"] = {
                "bytes_per_column_step": ALGO_BYTES_STEP, "achieved_GBps": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                "ms_per_step_events": ms_total,
                # the same tally charging a gated wrapper only for the columns it works on: albedo beyond init_timestep's
                # 500 written bytes only where the sun is up, bareground_fluxes beyond its 24 bytes only on bare columns,
                # canopy_fluxes beyond 32 bytes only on vegetated ones
                "active_bytes_per_column_step": round(
                    ALGO_BYTES_STEP - (1 - sun_frac) * (960 - 500) - veg_frac * (368 - 24) - (1 - veg_frac) * (2076 - 32), 1),
                "sunlit_fraction": round(sun_frac, 4), "vegetated_fraction": round(veg_frac, 4),
            }
            out["kernels"] = kern
        out["error_flags"] = flags
        out["device_state_GB"] = state_gb
        if other is not None:
            out["other_tier"] = other
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(host_state, workload=args.workload)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if D is not None:
        D.close()


if __name__ == "__main__":
    main()
