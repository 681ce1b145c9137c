"""One-wavefront-per-macro-cell kernel (csrc/small_wave.h, plane blocks b <= 48) against the oracle and against the two other routes of
the blocked family.  The cases walk through every shape the kernel is compiled for: 1 / 2 / 3 tiles per dimension, bordered arrow
(b + t <= 16 NT) or slab-form load rows, and every count of skipped k-slabs.  Sizes of the reference's own tests are among them:
2D elasticity 10 x 10 (test_integration_linear_elasticity.py:62-171), 3D Poisson 6^3 (test_integration_poisson.py:243-294)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-9

# (kind, dim, n)  ->  b, t, tiles, bordered
CASES = [
    ("poisson_matrix", 2, 12),    # b 12, t 2: NT 1, bordered, 3 of 4 slabs
    ("poisson_matrix", 2, 16),    # b 16, t 2: NT 1, slab-form load rows
    ("elasticity_voigt", 2, 7),   # b 14, t 3: NT 1, slab form
    ("poisson", 3, 4),            # b 16, t 3: NT 1, slab form, 9-point in-plane stencil
    ("poisson", 3, 3),            # b 9, t 3: NT 1, bordered, 3 of 4 slabs, smallest mesh the plan admits (stencil codes wrap around)
    ("elasticity", 2, 10),        # b 20, t 3: NT 2, bordered, 5 of 8 slabs
    ("poisson", 3, 5),            # b 25, t 3: NT 2, bordered, 7 of 8 slabs
    ("poisson_matrix", 2, 29),    # b 29, t 2: NT 2, bordered, 8 of 8 slabs
    ("elasticity", 2, 16),        # b 32, t 3: NT 2, slab form
    ("elasticity", 3, 3),         # b 27, t 6: NT 2, slab form
    ("poisson", 3, 6),            # b 36, t 3: NT 3, bordered, 9 of 12 slabs
    ("elasticity", 2, 21),        # b 42, t 3: NT 3, bordered, 11 of 12 slabs
    ("elasticity", 3, 4),         # b 48, t 6: NT 3, slab form
    ("poisson_matrix", 2, 47),    # b 47, t 2: NT 3, slab form (b + t = 49)
    ("poisson", 3, 7),            # b 49: LDS kernel (48 < b <= 64, csrc/small_fused.h), 13 of 16 k-slabs
    ("elasticity", 2, 28),        # b 56 on a 2D mesh: nested dissection since round 4 (register-resident fronts, csrc/mf_front_kernel.h)
    ("poisson", 3, 8),            # b 64: LDS kernel, all 16 k-slabs
]


def _inputs(p, kind, dim, nc, seed):
    rng = np.random.default_rng(seed)
    shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
    coef = rng.uniform(0.4, 2.5, size=shape)
    if kind == "poisson_matrix":
        coef[..., -1] = 0.2 * rng.uniform(-1, 1, size=shape[:2])
    if kind == "elasticity_voigt":  # SPD 3 x 3: diagonally dominant upper triangle (00, 01, 02, 11, 12, 22)
        coef[..., [1, 2, 4]] *= 0.1
    M = np.eye(dim)[None] + 0.2 * rng.standard_normal((nc, dim, dim))
    return coef, M


def _oracle_args(O, kind, dim, coef):
    """The oracle takes full tensors: A[n_el, d, d] for the matrix-valued Poisson kind, C[n_el, d, d, d, d] for the Voigt kind."""
    if kind == "poisson_matrix":
        pairs = [(0, 0), (1, 1), (0, 1)] if dim == 2 else [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
        A = np.zeros(coef.shape[:2] + (dim, dim))
        for q, (i, j) in enumerate(pairs):
            A[..., i, j] = A[..., j, i] = coef[..., q]
        return "poisson", A
    if kind == "elasticity_voigt":
        t = dim * (dim + 1) // 2
        iu = np.triu_indices(t)
        Cv = np.zeros(coef.shape[:2] + (t, t))
        Cv[..., iu[0], iu[1]] = coef
        Cv[..., iu[1], iu[0]] = coef
        E = O.unit_strains(dim)
        Ed = E * np.array([1.0 / np.sum(E[m] * E[m]) for m in range(t)])[:, None, None]  # dual basis: E^m : Ed^n = delta
        return "elasticity", np.einsum("cemn,mij,nkl->ceijkl", Cv, Ed, Ed)
    return kind, coef


def _plane_block(kind, dim, n):
    return (dim if kind.startswith("elasticity") else 1) * n ** (dim - 1)


@pytest.mark.parametrize("kind,dim,n", CASES)
def test_wave_kernel_matches_oracle(kind, dim, n):
    from hommx_amd import MicroCellPlan
    from oracle import hommx_oracle as O

    p = MicroCellPlan(dim, n, kind)
    b = _plane_block(kind, dim, n)
    assert p.kernel == ("small_wave" if b <= 48 else "multifrontal" if dim == 2 else "small_fused")
    coef, M = _inputs(p, kind, dim, 5, 3)
    A, info = p.solve(coef, M, return_info=True)
    assert not info.any()
    okind, ocoef = _oracle_args(O, kind, dim, coef)
    ref = O.effective_tensor_batch(okind, dim, n, ocoef[:2], M[:2])
    assert np.abs(A[:2] - ref).max() <= TOL * np.abs(ref).max()
    A0 = p.solve(coef)  # without the stratification matrix
    ref0 = O.effective_tensor_batch(okind, dim, n, ocoef[:1], None)
    assert np.abs(A0[:1] - ref0).max() <= TOL * np.abs(ref0).max()
    assert np.abs(A - np.swapaxes(A, 1, 2)).max() <= 1e-11 * np.abs(A).max()


def test_three_routes_agree(tmp_path):
    """wave kernel (default) == LDS-resident multi-wave kernel (HOMMX_SMALL_WAVES=2, csrc/small_fused.h) == HBM-resident kernels
    (HOMMX_NO_SMALL_FUSED=1) on the same inputs; the knobs are read when the plan is created, hence the child processes."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        from test_gpu_small_wave import CASES, _inputs
        for kind, dim, n in CASES:
            p = MicroCellPlan(dim, n, kind, flags=1)
            coef, M = _inputs(p, kind, dim, 70, 11)
            A, info = p.solve(coef, M, return_info=True)
            assert not info.any(), (kind, dim, n)
            np.save(sys.argv[1] + f"/{{kind}}_{{dim}}_{{n}}.npy", A)
        print("ok")
    """)
    outs = {}
    for tag, env_extra in (("wave", {}), ("lds", {"HOMMX_SMALL_WAVES": "2"}), ("hbm", {"HOMMX_NO_SMALL_FUSED": "1"})):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", code, str(d)], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
        outs[tag] = {c: np.load(d / f"{c[0]}_{c[1]}_{c[2]}.npy") for c in CASES}
    for c in CASES:
        scale = np.abs(outs["hbm"][c]).max()
        assert np.abs(outs["wave"][c] - outs["hbm"][c]).max() <= 1e-10 * scale, c
        assert np.abs(outs["lds"][c] - outs["hbm"][c]).max() <= 1e-10 * scale, c


def test_bad_cells_are_flagged_and_do_not_leak():
    """NaN and non-SPD coefficients: info[] > 0 for exactly those cells, their neighbours in the batch keep their bits."""
    from hommx_amd import MicroCellPlan

    for kind, dim, n in (("elasticity", 2, 10), ("poisson", 3, 6), ("poisson_matrix", 2, 16)):
        p = MicroCellPlan(dim, n, kind)
        coef, M = _inputs(p, kind, dim, 6, 5)
        good = p.solve(coef, M)
        bad = coef.copy()
        bad[1] = -bad[1]        # negative definite
        bad[4, 3] = np.nan
        A, info = p.solve(bad, M, return_info=True)
        assert info[1] > 0 and info[4] > 0 and not info[[0, 2, 3, 5]].any(), info
        assert np.array_equal(A[[0, 2, 3, 5]], good[[0, 2, 3, 5]])
        assert np.array_equal(p.solve(coef, M), good)  # bitwise reproducible
