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
    full = solve_sharded(_OraclePlan(), coef, M)
    mask = rng.uniform(size=coef.shape[1]) < 0.5
    values = rng.uniform(0.1, 3.0, size=(n_cells, 2))
    full2 = solve_sharded_two_phase(_OraclePlan(), mask, values, M)
    if rank == 0:
        q.put((full, full2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allgather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_cells = 7  # odd on purpose: ragged shards, one padded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_cells, q)) for r in range(2)]
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
