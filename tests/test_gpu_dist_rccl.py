"""The N>1 code path on the GPU with the RCCL backend (-m gpu): one rank (the GPU box has one card), i.e. the device-side
shard -> all_gather_into_tensor -> field sequence of hommx_amd/dist.py and bench.py, in a child process of its own."""

import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_solve_over_rccl_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np, torch, torch.distributed as dist
        from hommx_amd import MicroCellPlan
        from hommx_amd.dist import solve_sharded, all_gather_field, shard_range
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        rng = np.random.default_rng(1)
        p = MicroCellPlan(2, 16, "poisson")
        coef = rng.uniform(0.1, 3.0, size=(37, 512)); M = np.eye(2)[None] + 0.2 * rng.standard_normal((37, 2, 2))
        full, info = solve_sharded(p, coef, M, return_info=True)   # the shard stays on the device from the solve to the collective
        assert np.array_equal(full, p.solve(coef, M)) and info.dtype == np.int32 and not info.any()
        bad = coef.copy(); bad[5] = -1.0                           # a poisoned cell: its info flag travels with the field
        _, info = solve_sharded(p, bad, M, return_info=True)
        assert info[5] > 0 and (info != 0).sum() == 1
        from hommx_amd.dist import solve_sharded_two_phase
        mask = rng.uniform(size=512) < 0.4; vals = rng.uniform(0.1, 3.0, size=(37, 2))
        assert np.array_equal(solve_sharded_two_phase(p, mask, vals, M), p.solve_two_phase(mask, vals, M))
        out = torch.from_numpy(full).cuda()
        g = all_gather_field(out, 37)                         # device tensor in, device tensor out (bench.py's use)
        assert g.is_cuda and torch.equal(g, out)
        assert shard_range(37, 0, 1) == (0, 37, 37)
        dist.barrier(); dist.destroy_process_group(); print("ok")
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
