"""Multi-GPU sharding of the macro-cell loop (one process per GPU, torch.distributed over RCCL).

The reference shards the same way across MPI ranks: every rank solves the cell problems of the
macro cells it owns (/root/reference/src/hommx/hmm.py:307-310, docs/usage/usage.md:64-71) and the
results meet in PETSc's MatAssembly stash (hmm.py:325-330, :442).  Here the micro problems never
communicate; the only exchange is ONE all-gather of the effective-tensor field together with the
per-cell info flags (N_c x (t*t + 1) doubles, <= a few MB: latency-bound), so that every rank can
assemble the macro matrix and log failed cells exactly as a single rank would (hmm.py:320-323).

Every rank samples, uploads and solves ONLY its own block of cells; on the RCCL backend the shard
stays in device memory from the solve to the collective (one D2H copy of the gathered field).
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


_selected_device: int | None = None  # the caller's explicit choice (select_device); honoured even when it is ordinal 0


def select_device(device: int | None) -> None:
    """Record the device this rank's plans shall use (and make it torch's current device when torch is loaded and sees a GPU).
    ``torch.cuda.set_device(0)`` alone cannot be told from the untouched default, so a caller that MEANS ordinal 0 under a launcher
    whose LOCAL_RANK says otherwise calls this (or passes ``device=0`` to the solver class).  ``None`` forgets the choice."""
    import sys

    global _selected_device
    _selected_device = None if device is None else int(device)
    torch = sys.modules.get("torch")
    if device is not None and torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(int(device))


def _group_bound_device():
    """Device a process group was bound to with ``init_process_group(..., device_id=...)``: an explicit choice as well."""
    import sys

    dist = sys.modules.get("torch.distributed")
    try:
        if dist is None or not (dist.is_available() and dist.is_initialized()):
            return None
        pg = dist.distributed_c10d._get_default_group()
        dev = getattr(pg, "bound_device_id", None)
        if dev is not None and dev.type == "cuda" and dev.index is not None:
            return int(dev.index)
    except Exception:
        pass
    return None


def default_device(device_count=None) -> int:
    """Device ordinal a rank should use when no ``device=`` was passed.  Resolved LAZILY, at the first solve (a solver may be built
    before ``init_process_group`` / ``torch.cuda.set_device``).  In this order: the ordinal recorded by ``select_device`` (0
    included); the device the default process group is bound to (``device_id=``); the torch device the caller has already selected
    when it is not the untouched default 0; ``LOCAL_RANK % visible devices`` (a launcher that exposes one device per rank through
    HIP_VISIBLE_DEVICES leaves exactly one visible ordinal, 0); else 0."""
    import os
    import sys

    if _selected_device is not None:
        return _selected_device
    bound = _group_bound_device()
    if bound is not None:
        return bound
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        cur = int(torch.cuda.current_device())
        if cur != 0:
            return cur
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None:
        if device_count is None:
            from . import _lib

            device_count = _lib.load().hommx_device_count()
        return int(lr) % max(1, int(device_count))
    return 0


def all_gather_field(local, n_cells: int, group=None):
    """All-gather the per-rank shard of a field.

    ``local`` is a torch tensor [per_rank, ...] (padded shard) on this rank's device (RCCL) or on the
    CPU (gloo); returns the full field [n_cells, ...] on the same device on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = local.shape[0]
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:n_cells]


FAILED = -777.0  # failure flag of a rank in the packed buffer (its own row: never confused with a cell's info)


class ShardFailure(RuntimeError):
    """Raised on the ranks whose own block was fine when another rank's ``local_solve`` raised."""


def _unpad_index(n_cells: int, per: int, world: int) -> np.ndarray:
    """Positions of the real cells in the concatenation of the padded shards."""
    return np.concatenate([np.arange(r * per, r * per + max(0, min(n_cells, (r + 1) * per) - min(n_cells, r * per)))
                           for r in range(world)])


def run_sharded(t: int, n_cells: int, local_solve, group=None, device=None):
    """Solve this rank's block with ``local_solve(b, e) -> (A[e-b, t, t], info[e-b])`` and all-gather both.

    ``local_solve`` may return NumPy arrays or torch tensors (on the rank's GPU: the shard then never leaves the device before
    the collective).  The tensors and the info flags travel in ONE packed buffer [per_rank, t*t + 1] (info as a double: small
    integers are exact).  Returns NumPy ``(A_eff[n_cells, t, t], info[n_cells] int32)`` on every rank.
    """
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b, e, per = shard_range(n_cells, rank, world)
    nloc = e - b
    A = info = None
    failure = None
    if nloc > 0:
        try:
            A, info = local_solve(b, e)
        except Exception as exc:  # this rank still takes part in the collective; every rank raises after it (below)
            failure = exc
            A = info = None
    on_gpu = torch.is_tensor(A) and A.is_cuda
    if device is None:
        if on_gpu:
            device = A.device
        elif dist.get_backend(group) == "nccl":  # RCCL moves device memory only
            device = torch.device("cuda", default_device())
    elif not isinstance(device, torch.device):
        device = torch.device("cuda", int(device)) if dist.get_backend(group) == "nccl" else None
    # one extra row per rank carries the failure flag, so that ranks agree on an exception instead of waiting for ever
    buf = torch.zeros((per + 1, t * t + 1), dtype=torch.float64, device=device if device is not None else "cpu")
    if failure is not None:
        buf[per, t * t] = FAILED
    elif nloc > 0:
        At = A if torch.is_tensor(A) else torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64))
        it = info if torch.is_tensor(info) else torch.from_numpy(np.ascontiguousarray(info))
        buf[:nloc, : t * t] = At.reshape(nloc, t * t).to(buf.device)
        buf[:nloc, t * t] = it.to(buf.device, dtype=torch.float64)
    full = all_gather_field(buf, world * (per + 1), group).cpu().numpy().reshape(world, per + 1, t * t + 1)
    failed = [r for r in range(world) if full[r, per, t * t] == FAILED]
    if failed:
        if failure is not None:
            raise failure
        raise ShardFailure(f"micro-cell solve failed on rank(s) {failed} (see their exceptions); no field was assembled")
    full = full[:, :per].reshape(world * per, t * t + 1)[_unpad_index(n_cells, per, world)]
    return full[:, : t * t].reshape(n_cells, t, t).copy(), np.rint(full[:, t * t]).astype(np.int32)


def _has_device_entry(plan) -> bool:
    return hasattr(plan, "solve_device") and hasattr(plan, "device")


def _on_rccl(group=None) -> bool:
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"


def solve_block(plan, coef: np.ndarray, M: np.ndarray | None, group=None):
    """One rank's block through ``plan``: on a real plan under the RCCL backend the result stays on the device
    (torch tensors, torch's current stream); otherwise ``plan.solve`` on host arrays."""
    import torch

    if _has_device_entry(plan) and _on_rccl(group):
        dev = torch.device("cuda", plan.device)
        nc = coef.shape[0]
        c = torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64)).to(dev)
        m = None if M is None else torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64)).to(dev)
        out = torch.empty((nc, plan.t, plan.t), dtype=torch.float64, device=dev)
        info = torch.zeros(nc, dtype=torch.int32, device=dev)
        plan.solve_device(nc, c.data_ptr(), None if m is None else m.data_ptr(), out.data_ptr(), info.data_ptr(),
                          torch.cuda.current_stream(dev).cuda_stream)
        return out, info
    res = plan.solve(coef, M, return_info=True) if _accepts_return_info(plan.solve) else (plan.solve(coef, M), None)
    A, info = res
    return A, (np.zeros(len(coef), np.int32) if info is None else info)


def solve_block_separable(plan, family: str, table: np.ndarray, weights: np.ndarray | None, params: np.ndarray,
                          M: np.ndarray | None, group=None):
    """Separable coefficient (table of g once + (a, b) per cell): under RCCL the shard's tensors stay on the device."""
    import torch

    if hasattr(plan, "solve_separable_device") and hasattr(plan, "device") and _on_rccl(group):
        dev = torch.device("cuda", plan.device)
        nc = params.shape[0]
        tb = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(dev)
        nq = 1 if family == "affine" else int(table.shape[1])
        w = None if weights is None else torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float64)).to(dev)
        pr = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float64)).to(dev)
        m = None if M is None else torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64)).to(dev)
        out = torch.empty((nc, plan.t, plan.t), dtype=torch.float64, device=dev)
        info = torch.zeros(nc, dtype=torch.int32, device=dev)
        plan.solve_separable_device(nc, family, nq, tb.data_ptr(), None if w is None else w.data_ptr(), pr.data_ptr(),
                                    None if m is None else m.data_ptr(), out.data_ptr(), info.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream)
        return out, info
    return plan.solve_separable(family, table, weights, params, M, return_info=True)


def solve_block_two_phase(plan, mask: np.ndarray, values: np.ndarray, M: np.ndarray | None, group=None):
    import torch

    if hasattr(plan, "solve_two_phase_device") and hasattr(plan, "device") and _on_rccl(group):
        dev = torch.device("cuda", plan.device)
        nc = values.shape[0]
        mk = torch.from_numpy(np.ascontiguousarray(np.asarray(mask).astype(np.uint8))).to(dev)
        v = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(dev)
        m = None if M is None else torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64)).to(dev)
        out = torch.empty((nc, plan.t, plan.t), dtype=torch.float64, device=dev)
        info = torch.zeros(nc, dtype=torch.int32, device=dev)
        plan.solve_two_phase_device(nc, mk.data_ptr(), v.data_ptr(), None if m is None else m.data_ptr(), out.data_ptr(),
                                    info.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return out, info
    if _accepts_return_info(plan.solve_two_phase):
        return plan.solve_two_phase(mask, values, M, return_info=True)
    return plan.solve_two_phase(mask, values, M), np.zeros(len(values), np.int32)


def _accepts_return_info(fn) -> bool:
    import inspect

    try:
        return "return_info" in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


def solve_sharded(plan, coef: np.ndarray, M: np.ndarray | None, group=None, device=None, return_info: bool = False):
    """Convenience form for callers that hold the whole batch: every rank solves its block of ``coef`` / ``M`` and the field
    (and info) is all-gathered.  With no process group it degenerates to ``plan.solve``.  ``plan`` may be any object with
    ``.solve`` and ``.t`` (the gloo tests use the CPU oracle)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return plan.solve(coef, M, return_info=True) if return_info else plan.solve(coef, M)
    A, info = run_sharded(plan.t, coef.shape[0], lambda b, e: solve_block(plan, coef[b:e], None if M is None else M[b:e], group),
                          group, device if device is not None else getattr(plan, "device", None))
    return (A, info) if return_info else A


def solve_sharded_two_phase(plan, mask: np.ndarray, values: np.ndarray, M: np.ndarray | None, group=None, device=None,
                            return_info: bool = False):
    """Same for two-phase media: every rank holds the phase mask, the per-cell phase values shard."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return plan.solve_two_phase(mask, values, M, return_info=True) if return_info else plan.solve_two_phase(mask, values, M)
    A, info = run_sharded(plan.t, values.shape[0],
                          lambda b, e: solve_block_two_phase(plan, mask, values[b:e], None if M is None else M[b:e], group), group,
                          device if device is not None else getattr(plan, "device", None))
    return (A, info) if return_info else A
