"""C-ABI multi-GPU entry points (include/hommx_hip.h: hommx_comm_init_all / hommx_allgather_field / hommx_solve_batch_multi) on
the one-GPU box (-m gpu): a communicator of ONE device exercises the whole sequence -- shard upload, solve, pack [A | info],
RCCL all-gather (in place), unpack -- in a child process of its own (RCCL is dlopen'ed there, away from torch's copy)."""

import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_solve_batch_multi_one_device():
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import ctypes, numpy as np
        from hommx_amd import MicroCellPlan, _lib
        from hommx_amd.multi import MultiGpuSolver
        rng = np.random.default_rng(5)
        for kind, dim, n in (("poisson", 2, 16), ("elasticity", 3, 3)):
            ms = MultiGpuSolver(dim, n, kind, devices=[0])
            shape = (23, ms.n_el) + ((ms.n_comp,) if ms.n_comp > 1 else ())
            coef = rng.uniform(0.3, 3.0, size=shape)
            M = np.eye(dim)[None] + 0.2 * rng.standard_normal((23, dim, dim))
            ref = MicroCellPlan(dim, n, kind).solve(coef, M)
            A, info = ms.solve(coef, M, return_info=True)
            assert np.array_equal(A, ref) and not info.any()
            bad = coef.copy(); bad[7] = -1.0
            _, info = ms.solve(bad, M, return_info=True)
            assert info[7] > 0 and (info != 0).sum() == 1          # the info flag rides in the gathered buffer
            ms.close()
        # raw hommx_allgather_field, in place on one device (device memory through the HIP runtime the library already uses;
        # torch stays out of this process: it bundles its own RCCL)
        lib = _lib.load()
        hip = ctypes.CDLL(None)
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        h = ctypes.c_void_p()
        _lib.check(lib.hommx_comm_init_all(ctypes.byref(h), 1, None), "init")
        assert lib.hommx_comm_size(h) == 1
        src = np.arange(12, dtype=np.float64)
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), 96) == 0
        assert hip.hipMemcpy(d, src.ctypes.data, 96, 1) == 0
        ptrs = (ctypes.c_void_p * 1)(d.value)
        _lib.check(lib.hommx_allgather_field(h, ptrs, 12), "allgather")
        back = np.empty(12)
        assert hip.hipMemcpy(back.ctypes.data, d, 96, 2) == 0
        assert np.array_equal(back, src)
        # device-pointer form: the coefficient shard stays resident, the packed field [A | info] comes back gathered
        p2 = MicroCellPlan(2, 16, "poisson")
        coef = rng.uniform(0.3, 3.0, size=(11, p2.n_el)); coef[4] = -1.0
        ref, ref_info = p2.solve(coef, return_info=True)
        dc, dp = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dc), coef.nbytes) == 0 and hip.hipMalloc(ctypes.byref(dp), 11 * 5 * 8) == 0
        assert hip.hipMemcpy(dc, coef.ctypes.data, coef.nbytes, 1) == 0
        plans = (ctypes.c_void_p * 1)(p2._h.value)
        coefs, packs = (ctypes.c_void_p * 1)(dc.value), (ctypes.c_void_p * 1)(dp.value)
        _lib.check(lib.hommx_solve_batch_multi_device(h, plans, 11, coefs, None, packs), "multi_device")
        packed = np.empty((11, 5))
        assert hip.hipMemcpy(packed.ctypes.data, dp, packed.nbytes, 2) == 0
        A = np.empty((11, 2, 2)); info = np.empty(11, np.int32)
        _lib.check(lib.hommx_unpack_field(11, 1, 4, packed.ctypes.data, A.ctypes.data, info.ctypes.data), "unpack")
        assert np.array_equal(info, ref_info) and info[4] > 0
        ok = info == 0
        assert np.array_equal(A[ok], ref[ok])
        # ONE communicator reused with plans of another kind / dimension at the SAME cell count: the staging buffers are sized from
        # the plans (coefficients per cell, d x d, t*t + 1 per row), not only from the cells per device (advisor, round 3)
        for kind, dim, n in (("poisson", 2, 16), ("elasticity", 3, 3), ("poisson", 2, 8), ("elasticity", 3, 4)):
            pk = MicroCellPlan(dim, n, kind)
            shape = (23, pk.n_el) + ((pk.n_comp,) if pk.n_comp > 1 else ())
            coef = rng.uniform(0.3, 3.0, size=shape)
            M = np.eye(dim)[None] + 0.2 * rng.standard_normal((23, dim, dim))
            ref = pk.solve(coef, M)
            A = np.empty((23, pk.t, pk.t)); info = np.empty(23, np.int32)
            plans = (ctypes.c_void_p * 1)(pk._h.value)
            _lib.check(lib.hommx_solve_batch_multi(h, plans, 23, coef.ctypes.data, M.ctypes.data, A.ctypes.data, info.ctypes.data), "multi reuse")
            assert np.array_equal(A, ref) and not info.any(), (kind, dim, n)
        assert lib.hommx_comm_init_all(ctypes.byref(ctypes.c_void_p()), 2, (ctypes.c_int * 2)(0, 0)) == -1  # the same device twice
        lib.hommx_comm_destroy(h)
        print("ok")
    """)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
