"""The N > 1 path of bench.py on CPU: world_size-2 gloo, one process per rank.

What the multi-GPU run does besides launching kernels - block-split of the global column range, per-rank state
generation, barrier + max-over-ranks of the timed region, rank-0 aggregation - is exercised here with the oracle
standing in for the device (the product itself has no CPU path).  Checks that the shards tile the global
problem exactly and that per-rank results equal the corresponding slice of a single-process run (columns are
independent: no halo, no collective on the data path)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GLOBAL = 3001  # odd on purpose: ranks get different sizes
SEED = 31


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elmkernels_amd import decomp, synth
    from tests import helpers as H

    ft = H.field_table_from_oracle()
    start, count = decomp.block_range(N_GLOBAL, world, rank)
    cols, scal, soil = synth.make_state(ft, N_GLOBAL, tier="B", seed=SEED)  # global state, then this rank's slice
    mine = {k: v[start:start + count] for k, v in cols.items()}
    S = H.oracle_state(mine, scal, soil)
    dist.barrier()
    import time

    t0 = time.perf_counter()
    S.timestep7(1800.0)
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(el, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks
    # the one natural collective: global (min, max, sum) of the conservation diagnostics
    from elmkernels_amd import diagnostics

    S.soil_temperature(1800.0)
    S.surface_fluxes(1800.0)
    diag = S.evaluate_conservation(1800.0)
    gmms = diagnostics.global_min_max_sum(diagnostics.local_min_max_sum(diag))
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = count
    dist.all_reduce(sizes)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), start=start, count=count, t_veg=S["t_veg"], cgrnd=S["cgrnd"],
             snl=S["snl"], albd=S["albd"], elapsed=float(el.item()), sizes=sizes.numpy(), gmms=gmms)
    dist.destroy_process_group()


def test_two_rank_block_split(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from elmkernels_amd import synth
    from tests import helpers as H

    ft = H.field_table_from_oracle()
    cols, scal, soil = synth.make_state(ft, N_GLOBAL, tier="B", seed=SEED)
    S = H.oracle_state(cols, scal, soil)
    S.timestep7(1800.0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["count"]) for p in parts] == [1501, 1500]
    assert int(parts[0]["start"]) == 0 and int(parts[1]["start"]) == 1501
    assert parts[0]["sizes"].tolist() == [1501, 1500] and parts[0]["elapsed"] == parts[1]["elapsed"] > 0
    for name in ("t_veg", "cgrnd", "snl", "albd"):
        whole = np.concatenate([p[name] for p in parts])
        assert np.array_equal(whole, S[name], equal_nan=True), name
    # every rank holds the same global diagnostics, equal to those of the undivided domain
    from elmkernels_amd import diagnostics

    S.soil_temperature(1800.0)
    S.surface_fluxes(1800.0)
    ref = diagnostics.local_min_max_sum(S.evaluate_conservation(1800.0))
    assert np.array_equal(parts[0]["gmms"], parts[1]["gmms"])
    assert np.array_equal(parts[0]["gmms"][:, :2], ref[:, :2])  # min, max: exact
    assert np.allclose(parts[0]["gmms"][:, 2], ref[:, 2], rtol=1e-12, atol=1e-9)  # sum: association differs
