"""Global conservation diagnostics across ranks.

The physics path has no exchange step; the one natural collective is the reduction of the conservation
diagnostics (elmk_evaluate_conservation returns (min, max, sum) over a rank's columns) to the whole domain: MIN, MAX
and SUM all-reduces of an [8, 3] array - 192 bytes, off the timed path.  This mirrors the reference's min_max_sum
over MPI (src/utils/min_max_sum.hh:57-66) with torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def local_min_max_sum(per_column):
    """[ncols, k] -> [k, 3] (min, max, sum), the host-side form of what the device reduction returns."""
    a = np.asarray(per_column, dtype=np.float64)
    return np.stack([a.min(axis=0), a.max(axis=0), a.sum(axis=0)], axis=1)


def global_min_max_sum(mms, group=None, device=None):
    """All-reduce a rank-local [k, 3] (min, max, sum) array over the process group -> the global one on every rank."""
    import torch
    import torch.distributed as dist

    mms = np.ascontiguousarray(mms, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mms.copy()
    t = torch.from_numpy(mms.copy())
    if device is not None:
        t = t.to(device)
    mn, mx, sm = t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=group)
    return torch.stack([mn, mx, sm], dim=1).cpu().numpy()
