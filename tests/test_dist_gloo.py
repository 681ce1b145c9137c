"""N>1 path on CPU: world_size-2 gloo run of the shard + all-gather logic (hommx_amd/dist.py).

The GPU solver cannot run here, so each rank uses the CPU oracle as a stand-in for plan.solve; what is
under test is the partition (hmm.py:307-310: every rank solves the cells it owns) and the exchange.
"""

import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OraclePlan:
    t = 2

    def solve(self, coef, M=None):
        from oracle import hommx_oracle as O

        n = int(round(np.sqrt(coef.shape[1] / 2)))
        return O.effective_tensor_batch("poisson", 2, n, coef, M)

    def solve_two_phase(self, mask, values, M=None):
        return self.solve(np.where(np.asarray(mask, bool)[None, :], values[:, 1:2], values[:, 0:1]), M)


def _worker(rank, world, port, n_cells, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from hommx_amd.dist import solve_sharded, solve_sharded_two_phase

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    coef = rng.uniform(0.1, 3.0, size=(n_cells, 2 * 6 * 6))
    M = np.eye(2)[None] + 0.2 * rng.standard_normal((n_cells, 2, 2))
    full, info = solve_sharded(_OraclePlan(), coef, M, return_info=True)
    assert info.shape == (n_cells,) and not info.any()
    mask = rng.uniform(size=coef.shape[1]) < 0.5
    values = rng.uniform(0.1, 3.0, size=(n_cells, 2))
    full2 = solve_sharded_two_phase(_OraclePlan(), mask, values, M)
    if rank == 0:
        q.put((full, full2))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world,n_cells", [(2, 7), (8, 19)])  # ragged on purpose: 8 ranks x 3 cells, rank 6 owns one cell, rank 7 none
def test_two_rank_shard_and_allgather(world, n_cells):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cells, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, full2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(3)
    coef = rng.uniform(0.1, 3.0, size=(n_cells, 2 * 6 * 6))
    M = np.eye(2)[None] + 0.2 * rng.standard_normal((n_cells, 2, 2))
    ref = _OraclePlan().solve(coef, M)
    assert full.shape == ref.shape
    assert np.array_equal(full, ref)
    mask = rng.uniform(size=coef.shape[1]) < 0.5
    values = rng.uniform(0.1, 3.0, size=(n_cells, 2))
    assert np.array_equal(full2, _OraclePlan().solve_two_phase(mask, values, M))


# ---- end to end: PoissonHMM.solve() on two ranks (hmm.py:298-332 + :434-491 with the reference's MPI partition :307-310) ----------
class _CountingOraclePlan:
    """Stand-in for MicroCellPlan (oracle answers): records which cells this rank was asked to solve and flags one poisoned cell
    (non-positive coefficient) in info, as the kernels do (include/hommx_hip.h: info[c] > 0 = bad pivot)."""

    kind = "poisson"
    t = 2

    def __init__(self):
        self.batches = []

    def solve(self, coef, M=None, return_info=False, return_correctors=False):
        from oracle import hommx_oracle as O

        self.batches.append(coef.shape[0])
        n = int(round(np.sqrt(coef.shape[1] / 2)))
        info = (coef.min(axis=1) <= 0).astype(np.int32) * 3
        safe = np.where(coef > 0, coef, 1.0)
        A = O.effective_tensor_batch("poisson", 2, n, safe, M)
        A[info != 0] = np.nan
        return (A, info) if return_info else A


def _hmm_problem(poison_cell=None):
    from hommx_amd import hmm, mesh

    msh = mesh.create_unit_square(3, 3)  # 18 macro cells -> 9 per rank
    c = msh.cell_midpoints()

    def A(x, y):
        base = 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))
        if poison_cell is not None:  # make exactly one macro cell's coefficient non-positive
            base = np.where(np.isclose(x[0], c[poison_cell, 0]) & np.isclose(x[1], c[poison_cell, 1]), -1.0, base)
        return base

    return hmm.PoissonHMM(msh, A, lambda x: 1.0 + x[0], mesh.create_unit_square(6, 6), 0.01)


def _hmm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import logging

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for tag, poison in (("clean", None), ("poisoned", 13)):
        h = _hmm_problem(poison)
        plan = _CountingOraclePlan()
        h._plan = plan
        errors = []
        handler = logging.Handler()
        handler.emit = lambda rec, errors=errors: errors.append(rec.getMessage()) if rec.levelno >= logging.ERROR else None
        logging.getLogger("hommx_amd.hmm").addHandler(handler)
        u = h.solve()
        logging.getLogger("hommx_amd.hmm").removeHandler(handler)
        out[tag] = dict(u=u.x.array.copy(), info=h.cell_info.copy(), batches=list(plan.batches), errors=errors,
                        AH=h.effective_tensors.copy(), qdeg=h.quadrature_degree_used)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hmm_solve_end_to_end():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hmm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-rank answer in this process (no process group here)
    h = _hmm_problem(None)
    h._plan = _CountingOraclePlan()
    u_ref = h.solve().x.array
    assert h.quadrature_degree_used == 3  # smooth coefficient: the default policy picks UFL's estimate
    for r in (0, 1):
        clean = got[r]["clean"]
        assert clean["qdeg"] == 3
        assert clean["batches"] == [9]                      # every rank sampled and solved ONLY its 9 cells
        assert np.array_equal(clean["AH"], h.effective_tensors)
        assert np.array_equal(clean["u"], u_ref)             # macro solution identical to the single-rank one, on both ranks
        assert not clean["info"].any() and not clean["errors"]
        bad = got[r]["poisoned"]
        assert bad["info"][13] == 3 and bad["info"].sum() == 3   # the poisoned cell's flag survives the all-gather on both ranks
        assert any("cell 13" in m for m in bad["errors"])        # and is logged like the reference (hmm.py:320-323)


# ---- a failing rank must not leave the others in the collective for ever --------------------------------------------------------
def _failing_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from hommx_amd.dist import ShardFailure, run_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local(b, e):
        if rank == 1:
            raise ValueError("matrix-valued A must be symmetric (only this rank's shard trips the check)")
        return np.ones((e - b, 2, 2)), np.zeros(e - b, np.int32)

    try:
        run_sharded(2, 7, local)
        q.put((rank, "returned"))
    except ValueError as exc:
        q.put((rank, "own:" + str(exc)[:20]))
    except ShardFailure as exc:
        q.put((rank, "peer:" + str(exc)))
    # the group is still usable afterwards: both ranks left the collective
    A, info = run_sharded(2, 7, lambda b, e: (np.full((e - b, 2, 2), float(rank)), np.zeros(e - b, np.int32)))
    assert A[0, 0, 0] == 0.0 and A[-1, 0, 0] == 1.0
    dist.barrier()
    dist.destroy_process_group()


def test_failure_on_one_rank_raises_on_all_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1].startswith("own:matrix-valued")
    assert got[0].startswith("peer:") and "rank(s) [1]" in got[0]


def test_default_device_is_resolved_lazily(monkeypatch):
    """A solver built BEFORE init_process_group must not freeze device 0 on every rank: the device is chosen at the first solve
    from LOCAL_RANK modulo the visible devices (one device per rank through HIP_VISIBLE_DEVICES -> ordinal 0)."""
    sys.path.insert(0, ROOT)
    from hommx_amd import hmm, mesh
    from hommx_amd.dist import default_device

    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert default_device(device_count=8) == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert default_device(device_count=8) == 3
    assert default_device(device_count=1) == 0   # launcher exposed ONE device to this rank
    assert default_device(device_count=0) == 0   # no device: the plan constructor reports that, not an index error here
    h = hmm.PoissonHMM(mesh.create_unit_square(2, 2), lambda x, y: 1.0 + 0 * y[0], lambda x: 1.0, mesh.create_unit_square(4, 4), 0.1)
    assert h._device is None                      # nothing decided at construction time
    h2 = hmm.PoissonHMM(mesh.create_unit_square(2, 2), lambda x, y: 1.0 + 0 * y[0], lambda x: 1.0, mesh.create_unit_square(4, 4), 0.1, device=5)
    assert h2._device == 5                        # an explicit device always wins


def test_explicit_device_zero_is_honoured(monkeypatch):
    """`current_device() == 0` used to read as "not chosen", so an explicit choice of ordinal 0 under
    LOCAL_RANK=3 was overridden.  The caller's choice is now tracked (dist.select_device), 0 included."""
    sys.path.insert(0, ROOT)
    from hommx_amd import dist as D

    monkeypatch.setenv("LOCAL_RANK", "3")
    try:
        assert D.default_device(device_count=8) == 3
        D.select_device(0)
        assert D.default_device(device_count=8) == 0
        D.select_device(5)
        assert D.default_device(device_count=8) == 5
        D.select_device(None)
        assert D.default_device(device_count=8) == 3
    finally:
        D.select_device(None)
