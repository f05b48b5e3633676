"""CPU stand-in for bench.build_state (tests only, selected through ELMK_BENCH_REHEARSAL=tests.rehearsal:make_state).

bench.py's multi-rank control flow - rank discovery / self-spawn, the reference's 1-D block split, barrier,
max-over-ranks of the timed region, rank-0 aggregation and the single JSON line - is device-independent; the one thing
that needs a GPU is the step itself.  This factory replaces exactly that: the state is an OracleState holding this rank's
block of a GLOBAL synthetic state (so that the test can check that the shards tile the undivided problem), and the step
is the oracle's timestep7.  The product has no CPU path; nothing under elmkernels_amd/ or in bench.py's GPU path imports
this module."""
import os

import numpy as np

N_SEED = 31


class RehearsalState:
    def __init__(self, ncols, tier):
        from elmkernels_amd import decomp, synth
        from tests import helpers as H

        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        nglobal = int(os.environ["ELMK_REHEARSAL_NGLOBAL"])
        ft = H.field_table_from_oracle()
        start, count = decomp.block_range(nglobal, world, rank)
        assert count == ncols, (count, ncols)
        cols, scal, soil = synth.make_state(ft, nglobal, tier=tier, seed=N_SEED)
        mine = {k: v[start:start + count] for k, v in cols.items()}
        self.S = H.oracle_state(mine, scal, soil)
        self.start, self.count, self.rank = start, count, rank
        self.steps = 0
        self.base = self.S.clone()

    def rehearsal_step(self):
        # test hook: one rank dies before the barrier (tests/test_sharding_gloo.py checks that the launcher ends the others)
        if os.environ.get("ELMK_REHEARSAL_DIE_RANK") == str(self.rank):
            os._exit(7)
        self.S.copy_from(self.base)  # every step starts from the same state: the result is that of ONE timestep
        self.S.timestep7(1800.0)
        self.steps += 1

    def __getitem__(self, name):
        return self.S[name]

    def sync(self):
        pass

    def error_summary(self):
        return int(np.bitwise_or.reduce(self.S["err_flags"])) if self.count else 0, -1

    @property
    def device_bytes(self):
        return 0

    def close(self):
        out = os.environ.get("ELMK_REHEARSAL_OUT")
        if out and self.S is not None:
            np.savez(os.path.join(out, f"rank{self.rank}.npz"), start=self.start, count=self.count, steps=self.steps,
                     t_veg=self.S["t_veg"], cgrnd=self.S["cgrnd"], snl=self.S["snl"], albd=self.S["albd"])
        self.S = None


def make_state(ncols, device, tier, seed):
    return RehearsalState(ncols, tier), None
