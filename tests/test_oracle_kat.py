"""The oracle against every known answer the reference's own tests hold -- CPU only."""

import numpy as np
import pytest

from oracle import hommx_oracle as O


def _AH(A, n, x=(0.0, 0.0), degree=0, M=None, dim=2):
    coef = O.sample_coefficient(A, np.asarray(x, float), dim, n, degree)
    return O.effective_tensor(O.build_cell_problem("poisson", dim, n, coef, M))


def test_periodic_topology_2d():
    """test/unit/test_unit.py:25-54: only boundary dofs are slaves, (1,1)->(0,0), pairs differ by unit vectors."""
    n = 5
    x, _ = O.unit_cell_mesh(2, n)
    slaves, masters = O.periodic_slaves_masters(2, n)
    assert np.all(np.isclose(x[slaves], 1.0).any(axis=1))
    assert len(slaves) == 2 * n + 1
    corner = np.nonzero(np.all(np.isclose(x[slaves], 1.0), axis=1))[0]
    assert len(corner) == 1 and np.allclose(x[masters[corner[0]]], 0.0)
    diff = x[slaves] - x[masters]
    assert np.all((np.isclose(diff, 0) | np.isclose(diff, 1)))
    assert np.all(diff.sum(axis=1) >= 1)
    # masters are never slaves; the map is the torus map used by the assembly
    assert not set(masters) & set(slaves)
    pm = O.periodic_master_map(2, n)
    g = np.rint(x * n).astype(int) % n
    assert np.array_equal(pm, g[:, 0] + n * g[:, 1])
    assert np.array_equal(pm[slaves], pm[masters])


def test_periodic_topology_3d():
    """test/unit/test_unit.py:57-103."""
    n = 3
    x, _ = O.unit_cell_mesh(3, n)
    slaves, masters = O.periodic_slaves_masters(3, n)
    assert len(slaves) == (n + 1) ** 3 - n**3
    diff = x[slaves] - x[masters]
    assert np.all((np.isclose(diff, 0) | np.isclose(diff, 1)))
    assert not set(masters) & set(slaves)
    assert len(np.unique(O.periodic_master_map(3, n))) == n**3


def test_mesh_is_a_partition():
    for dim, n in ((2, 4), (3, 3)):
        x, cells = O.unit_cell_mesh(dim, n)
        X = x[cells]
        vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / (2 if dim == 2 else 6)
        assert np.isclose(vol.sum(), 1.0) and np.all(vol > 0)


def test_analytical_example_1():
    """test_integration_poisson.py:121-143: A = 1/(2+cos 2 pi y0) => A_H = diag(1/2, 1/sqrt 3) (O(h^2))."""
    A = lambda x, y: 1.0 / (2.0 + np.cos(2 * np.pi * y[0]))
    errs = []
    for n in (15, 30, 60):
        AH = _AH(A, n, degree=3)
        assert abs(AH[0, 1]) < 1e-12 and abs(AH[1, 1] - 1 / np.sqrt(3)) < 1e-9
        errs.append(abs(AH[0, 0] - 0.5))
    assert errs[0] < 1.1e-3 and errs[1] < errs[0] / 3.5 and errs[2] < errs[1] / 3.5  # second order


def test_analytical_example_2():
    """test_integration_poisson.py:146-185: A_H(x) = diag(sqrt(a^2-0.15^2), a), a = 0.33+0.15 sin 2 pi x0."""
    A = lambda x, y: 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))
    for x0 in (0.1, 0.4, 0.8):
        a = 0.33 + 0.15 * np.sin(2 * np.pi * x0)
        AH = _AH(A, 60, x=(x0, 0.3), degree=3)
        assert abs(AH[0, 0] - np.sqrt(a * a - 0.15**2)) < 2e-4
        assert abs(AH[1, 1] - a) < 1e-6
        assert abs(AH[0, 1]) < 1e-12


def test_laminate_exact():
    """laminate.py:101-102 with 4 | n and the centroid rule: P1 is exact => harmonic / arithmetic mean to round-off."""
    A = lambda x, y: np.where(np.cos(2 * np.pi * y[0]) < 0, 5.0, 0.05)
    for n in (4, 16, 32):
        AH = _AH(A, n)
        assert np.allclose(AH, np.diag([0.09900990099009901, 2.525]), rtol=0, atol=1e-13)


def test_stratified_laminate_closed_form(rng):
    A = lambda x, y: np.where(np.cos(2 * np.pi * y[1]) < 0, 5.0, 0.05)
    M = np.eye(2) + 0.5 * rng.standard_normal((2, 2))
    AH = _AH(A, 16, M=M)
    m = M[:, 1]
    P = np.outer(m, m) / (m @ m)
    assert np.allclose(AH, 2.525 * (np.eye(2) - P) + 0.09900990099009901 * P, atol=1e-12)


def test_hmm_local_stiffness_equals_compact_form(rng):
    """test_integration_poisson.py:188-240 pins S_loc == vol(T) G A_hom G^T; here: the literal per-basis-function
    algorithm of hmm.py:334-369 equals the compact form, for all four classes."""
    X2 = np.array([[0.1, 0.2], [0.4, 0.25], [0.2, 0.6]])
    for kind, dim, n in (("poisson", 2, 6), ("elasticity", 2, 5), ("poisson", 3, 3), ("elasticity", 3, 3)):
        n_el = (2 if dim == 2 else 6) * n**dim
        coef = rng.uniform(0.1, 5.0, n_el) if kind == "poisson" else np.stack(
            [rng.uniform(1, 2, n_el), rng.uniform(0.5, 50, n_el)], axis=1)
        X = X2 if dim == 2 else rng.uniform(0, 1, (4, 3))
        for M in (None, np.eye(dim) + 0.3 * rng.standard_normal((dim, dim))):
            cp = O.build_cell_problem(kind, dim, n, coef, M)
            AH = O.effective_tensor(cp)
            S1 = O.local_stiffness_from_tensor(kind, X, AH)
            for eps in (0.5, 1e-4):
                S2 = O.local_stiffness_reference_shaped(kind, dim, n, X, coef, eps, M)
                assert np.abs(S1 - S2).max() < 1e-10 * np.abs(S1).max()
            assert np.abs(AH - O.effective_tensor(cp, form="schur")).max() < 1e-11 * np.abs(AH).max()


def test_constant_hooke_tensor_is_reproduced():
    """test_integration_linear_elasticity.py:205-322: constant A => C_H = C (HMM matrix == plain FEM matrix)."""
    for dim, n in ((2, 4), (3, 3)):
        n_el = (2 if dim == 2 else 6) * n**dim
        cp = O.build_cell_problem("elasticity", dim, n, np.tile([1.25, 1.0], (n_el, 1)))
        CH = O.effective_tensor(cp)
        C = O.isotropic_hooke(1.25, 1.0, dim)
        E = O.unit_strains(dim)
        assert np.abs(CH - np.einsum("mij,ijkl,nkl->mn", E, C, E)).max() < 1e-13
        # kernel of K = translations
        tr = np.tile(np.eye(dim), (n**dim, 1))
        assert np.abs(cp.K @ tr).max() < 1e-13


def test_layered_elasticity_exact():
    """layers normal to y0, lambda=1, mu in {5, 0.5}: C_H[00,00] = 1/<1/(lam+2mu)>, shear 1/<1/mu>."""
    n = 4
    x, cells = O.unit_cell_mesh(3, n)
    yb = x[cells].mean(axis=1)
    mu = np.where(np.cos(2 * np.pi * yb[:, 0]) < 0, 5.0, 0.5)
    cp = O.build_cell_problem("elasticity", 3, n, np.stack([np.ones_like(mu), mu], axis=1))
    CH = O.effective_tensor(cp)
    assert abs(CH[0, 0] - 1.0 / np.mean([1 / 11.0, 1 / 2.0])) < 1e-12
    # E^{01} : C_H : E^{01} with tensorial unit strain = mu_eff  (sigma_01 = 2 mu e_01, energy 2*mu*(1/2)^2*2)
    assert abs(CH[3, 3] - 1.0 / np.mean([1 / 5.0, 1 / 0.5])) < 1e-12


@pytest.mark.parametrize("dim", [2, 3])
def test_quadrature_rules(dim):
    for deg in (0, 2, 3):
        p, w = O.quadrature_rule(dim, deg)
        assert np.isclose(w.sum(), 1) and np.allclose(p.sum(axis=1), 1)
        # exactness on monomials of barycentric coordinate 0: int l0^k = k! d! / (k+d)!
        from math import factorial as f

        for k in range(deg + 1):
            assert np.isclose(w @ p[:, 0] ** k, f(k) * f(dim) / f(k + dim))
