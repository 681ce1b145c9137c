"""Multi-GPU sharding of the macro-cell loop (one process per GPU, torch.distributed over RCCL).

The reference shards the same way across MPI ranks: every rank solves the cell problems of the
macro cells it owns (/root/reference/src/hommx/hmm.py:307-310, docs/usage/usage.md:64-71) and the
results meet in PETSc's MatAssembly stash (hmm.py:325-330, :442).  Here the micro problems never
communicate; the only exchange is ONE all-gather of the effective-tensor field (N_c x t x t doubles,
<= a few MB: latency-bound), so that every rank can assemble the macro matrix.
"""

from __future__ import annotations

import numpy as np


def shard_range(n_cells: int, rank: int, world: int) -> tuple[int, int, int]:
    """Contiguous block partition padded to equal counts: returns (begin, end, per_rank).

    cells_r = [r * ceil(N_c / P), min(N_c, (r+1) * ceil(N_c / P)))  (SURVEY 8(e))."""
    per = -(-n_cells // world)
    b = min(n_cells, rank * per)
    e = min(n_cells, (rank + 1) * per)
    return b, e, per


def all_gather_field(local, n_cells: int, group=None):
    """All-gather the per-rank shard of the effective-tensor field.

    ``local`` is a torch tensor [per_rank, t, t] (padded shard) on this rank's device (RCCL) or on the
    CPU (gloo); returns the full field [n_cells, t, t] on the same device on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = local.shape[0]
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:n_cells]


def _gather_shards(plan, n_cells: int, solve_range, group=None, device=None):
    """Run ``solve_range(b, e)`` on this rank's block of cells and all-gather the field over the group's backend."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b, e, per = shard_range(n_cells, rank, world)
    t = plan.t
    local = np.zeros((per, t, t))
    if e > b:
        local[: e - b] = solve_range(b, e)
    tl = torch.from_numpy(local)
    if device is None and dist.get_backend(group) == "nccl":  # RCCL moves device memory only
        device = torch.device("cuda", getattr(plan, "device", torch.cuda.current_device()))
    if device is not None:
        tl = tl.to(device)
    full = all_gather_field(tl, world * per, group)
    # padded shards sit at the tail of every rank's block; strip them
    idx = np.concatenate([np.arange(r * per, r * per + max(0, min(n_cells, (r + 1) * per) - min(n_cells, r * per)))
                          for r in range(world)])
    return full.cpu().numpy()[idx]


def solve_sharded(plan, coef: np.ndarray, M: np.ndarray | None, group=None, device=None):
    """Solve this rank's block of macro cells and all-gather the field (host-array convenience path).

    With an initialised process group the exchange runs over the group's backend (``nccl`` = RCCL on
    the GPUs, ``gloo`` in the CPU tests, where ``plan`` may be any object with ``.solve`` and ``.t``).
    Without one it degenerates to ``plan.solve``.
    """
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return plan.solve(coef, M)
    return _gather_shards(plan, coef.shape[0], lambda b, e: plan.solve(coef[b:e], None if M is None else M[b:e]), group, device)


def solve_sharded_two_phase(plan, mask: np.ndarray, values: np.ndarray, M: np.ndarray | None, group=None, device=None):
    """Same for two-phase media (``plan.solve_two_phase``): every rank holds the phase mask, the per-cell phase values shard."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return plan.solve_two_phase(mask, values, M)
    return _gather_shards(plan, values.shape[0],
                          lambda b, e: plan.solve_two_phase(mask, values[b:e], None if M is None else M[b:e]), group, device)
