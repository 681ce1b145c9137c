"""API robustness of the C-ABI library on the GPU (-m gpu): batch-size changes, several plans, misuse, chunked workspaces."""

import ctypes as C
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_size_changes_and_independence_of_cells(rng):
    """A cell's result does not depend on the batch it travels in (fused: bit for bit), staging buffers regrow."""
    from hommx_amd import MicroCellPlan

    p = MicroCellPlan(2, 32, "poisson")
    coef = rng.uniform(0.05, 5.0, size=(1500, 2048))
    M = np.eye(2)[None] + 0.2 * rng.standard_normal((1500, 2, 2))
    full = p.solve(coef, M)
    for lo, hi in ((0, 1), (1, 8), (8, 72), (72, 1072), (1497, 1500), (0, 1500)):
        assert np.array_equal(p.solve(coef[lo:hi], M[lo:hi]), full[lo:hi])
    q = MicroCellPlan(3, 4, "elasticity")
    c3 = rng.uniform(0.5, 2.0, size=(40, q.n_el, 2))
    f3 = q.solve(c3)
    for lo, hi in ((0, 1), (1, 5), (5, 40)):
        assert np.allclose(q.solve(c3[lo:hi]), f3[lo:hi], rtol=1e-12, atol=0)


def test_plans_coexist_and_recycle(rng):
    from hommx_amd import MicroCellPlan

    coef2 = rng.uniform(0.1, 2.0, size=(9, 512))
    ref = MicroCellPlan(2, 16, "poisson").solve(coef2)
    plans = [MicroCellPlan(2, 16, "poisson"), MicroCellPlan(2, 16, "poisson", flags=1), MicroCellPlan(3, 3, "poisson")]
    c3 = rng.uniform(0.1, 2.0, size=(4, plans[2].n_el))
    r3 = plans[2].solve(c3)
    for _ in range(3):  # interleaved use
        assert np.array_equal(plans[0].solve(coef2), ref)
        assert np.allclose(plans[1].solve(coef2), ref, rtol=1e-11)
        assert np.array_equal(plans[2].solve(c3), r3)
    for _ in range(40):  # create / destroy cycles
        assert np.array_equal(MicroCellPlan(2, 16, "poisson").solve(coef2[:2]), ref[:2])


def test_null_and_invalid_arguments_are_rejected_not_dereferenced():
    from hommx_amd import _lib

    lib = _lib.load()
    plan = C.c_void_p()
    assert lib.hommx_plan_create(C.byref(plan), None) == -1  # HOMMX_EINVAL
    assert lib.hommx_plan_create(None, C.byref(_lib.PlanDesc(2, 16, 0, 0, 0))) == -1
    for bad in (_lib.PlanDesc(4, 16, 0, 0, 0), _lib.PlanDesc(2, 16, 9, 0, 0), _lib.PlanDesc(2, 0, 0, 0, 0)):
        assert lib.hommx_plan_create(C.byref(plan), C.byref(bad)) == -1
        assert lib.hommx_last_error()
    assert lib.hommx_plan_create(C.byref(plan), C.byref(_lib.PlanDesc(2, 16, 0, 999, 0))) in (-1, -3)  # no such device
    assert lib.hommx_plan_create(C.byref(plan), C.byref(_lib.PlanDesc(2, 16, 0, 0, 0))) == 0
    out = np.zeros((2, 2, 2))
    coef = np.ones((2, 512))
    dp = C.POINTER(C.c_double)
    assert lib.hommx_solve_batch(None, 2, coef.ctypes.data_as(dp), None, out.ctypes.data_as(dp), None) == -1
    assert lib.hommx_solve_batch(plan, 2, None, None, out.ctypes.data_as(dp), None) == -1
    assert lib.hommx_solve_batch(plan, 2, coef.ctypes.data_as(dp), None, None, None) == -1
    assert lib.hommx_solve_batch(plan, -1, coef.ctypes.data_as(dp), None, out.ctypes.data_as(dp), None) == -1
    assert lib.hommx_solve_batch(plan, 2, coef.ctypes.data_as(dp), None, out.ctypes.data_as(dp), None) == 0  # info optional
    assert np.allclose(out, np.eye(2))
    assert lib.hommx_plan_destroy(plan) == 0
    assert lib.hommx_plan_destroy(None) == 0


def test_blocked_workspace_in_several_chunks(tmp_path):
    """A 1 MB workspace budget forces the blocked path through many equalised chunks: same tensors as in one chunk."""
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        rng = np.random.default_rng(3)
        p = MicroCellPlan(3, 4, "elasticity")
        coef = rng.uniform(0.5, 2.0, size=(37, p.n_el, 2))
        A, chi, info = p.solve(coef, return_info=True, return_correctors=True)
        np.savez({str(tmp_path / 'out.npz')!r}, A=A, chi=chi, info=info)
    """)
    outs = []
    for budget in ("0.001", "8"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HOMMX_BLOCKED_MEM_GB=budget), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(tmp_path / "out.npz")))
    assert not outs[0]["info"].any()
    assert np.allclose(outs[0]["A"], outs[1]["A"], rtol=1e-12, atol=0)
    assert np.allclose(outs[0]["chi"], outs[1]["chi"], rtol=1e-9, atol=1e-13)


def test_plan_reserve_then_solve(rng):
    """hommx_plan_reserve allocates the route's workspace up front; solves afterwards give the same bits as without it."""
    from hommx_amd import MicroCellPlan, _lib

    for dim, n, kind in ((3, 6, "elasticity"), (3, 6, "poisson"), (2, 16, "poisson")):
        a, b = MicroCellPlan(dim, n, kind), MicroCellPlan(dim, n, kind)
        b.reserve(50)
        shape = (17, a.n_el) + ((a.n_comp,) if a.n_comp > 1 else ())
        coef = rng.uniform(0.5, 2.0, size=shape)
        assert np.array_equal(a.solve(coef), b.solve(coef))
    with pytest.raises(_lib.HommxLibraryError):
        a.reserve(-1)
