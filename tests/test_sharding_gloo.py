"""The N > 1 path of bench.py on CPU: world_size-2 gloo, one process per rank.

test_bench_main_*: bench.py's OWN main() - self-spawn of the rank processes (`python bench.py --gpus 2`), and the
driver's `python -m torch.distributed.run ... bench.py --gpus 2` - with only the device step replaced (tests/rehearsal.py:
the oracle on this rank's block).  Checks the JSON line (n_gpus, columns_total, weak scaling), that the shards are the
reference's 1-D block split (src/utils/utils.cc:27-44) and that they tile the undivided problem exactly.

test_two_rank_block_split: the split arithmetic and the one natural collective (global min / max / sum of the
conservation diagnostics) under gloo."""
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


def _single_process_reference(n):
    sys.path.insert(0, ROOT)
    from elmkernels_amd import synth
    from tests import helpers as H
    from tests import rehearsal

    ft = H.field_table_from_oracle()
    cols, scal, soil = synth.make_state(ft, n, tier="B", seed=rehearsal.N_SEED)
    S = H.oracle_state(cols, scal, soil)
    S.timestep7(1800.0)
    return S


def _run_bench(cmd, tmp_path, nglobal):
    import json
    import subprocess

    env = dict(os.environ, ELMK_BENCH_REHEARSAL="tests.rehearsal:make_state", ELMK_REHEARSAL_OUT=str(tmp_path),
               ELMK_REHEARSAL_NGLOBAL=str(nglobal), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_shards(tmp_path, line, per_rank, world, steps, warmup):
    nglobal = per_rank * world
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["steps"] == steps and line["warmup"] == warmup
    assert line["config"]["columns_total"] == nglobal and line["config"]["columns_per_gpu"] == per_rank
    assert line["metric"] == "gridcell-timesteps/sec" and line["value"] > 0
    assert abs(line["value"] - nglobal * steps / (line["ms_per_step"] * 1e-3 * steps)) < 1e-6 * line["value"]
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    from elmkernels_amd import decomp

    assert [(int(p["start"]), int(p["count"])) for p in parts] == decomp.all_ranges(nglobal, world)
    assert all(int(p["steps"]) == steps + warmup for p in parts)  # W untimed + exactly K timed steps on every rank
    S = _single_process_reference(nglobal)
    for name in ("t_veg", "cgrnd", "snl", "albd"):
        whole = np.concatenate([p[name] for p in parts])
        assert np.array_equal(whole, S[name], equal_nan=True), name


def test_bench_main_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent starts the two rank processes itself."""
    line = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--cols", "700", "--steps", "2", "--warmup", "1",
                       "--tier", "B", "--no-cpu-baseline"], tmp_path, 1400)
    _check_shards(tmp_path, line, 700, 2, 2, 1)
    # every rank's own time is in the line (a straggler would show), the job's time is their maximum
    assert len(line["per_rank_ms"]) == 2 and abs(max(line["per_rank_ms"]) - line["ms_per_step"]) < 1e-3 * line["ms_per_step"] + 1e-3


def test_a_rank_that_dies_early_ends_the_run_within_seconds(tmp_path):
    """One rank exits before the barrier: the parent must stop the other rank (which is waiting in the rendezvous or in the
    barrier) and return non-zero quickly, not after the backend's timeout."""
    import subprocess
    import time

    env = dict(os.environ, ELMK_BENCH_REHEARSAL="tests.rehearsal:make_state", ELMK_REHEARSAL_OUT=str(tmp_path),
               ELMK_REHEARSAL_NGLOBAL="400", ELMK_REHEARSAL_DIE_RANK="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--cols", "200", "--steps", "2", "--warmup", "1", "--tier", "B",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    took = time.time() - t0
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-1500:])
    assert "rank 1 exited with 7" in r.stderr.decode()
    assert took < 120, took  # (importing torch twice takes most of it; the gloo timeout would be 30 minutes)
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]  # no result line from a broken run


def test_bench_main_under_torch_distributed_run(tmp_path):
    """The driver's launch line for N > 1."""
    port = 20000 + (os.getpid() % 20000)
    line = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                       "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--cols", "333",
                       "--steps", "2", "--warmup", "1", "--tier", "B"], tmp_path, 666)
    _check_shards(tmp_path, line, 333, 2, 2, 1)


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
