"""On-device sampling of separable coefficients (include/hommx_hip.h hommx_solve_batch_separable):
the smooth coefficient families of the reference's own tests (test_integration_poisson.py:124-125, 149-150, 197, 268)
== the host-sampled element stream, bit for bit (-m gpu)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(dim, n, degree=3):
    from hommx_amd import hmm, mesh

    micro = mesh.create_unit_square(n, n) if dim == 2 else mesh.create_unit_cube(n, n, n)
    bary, w = hmm.micro_quadrature(dim, degree)
    return np.einsum("qa,eak->eqk", bary, micro.cell_vertices()), w


@pytest.mark.parametrize("family", ["affine", "reciprocal"])
@pytest.mark.parametrize("dim,n", [(2, 32), (2, 15), (2, 7), (3, 6)])
def test_separable_equals_host_stream_bitwise(family, dim, n, rng):
    """Fused 2D kernel (n <= 32) and the blocked family (3D) against plan.solve of the stream the host forms with the same formula."""
    from hommx_amd import MicroCellPlan, hmm

    yq, w = _tables(dim, n)
    g = (lambda y: np.cos(2 * np.pi * y[0])) if family == "reciprocal" else (lambda y: np.sin(2 * np.pi * y[0]) * np.cos(2 * np.pi * y[1]))
    co = hmm.Separable(family, lambda x: 2.0 + x[0], lambda x: 0.5 + 0.3 * x[1], g)
    nc = 6
    c = np.concatenate([rng.uniform(size=(nc, 2)), np.zeros((nc, 1))], axis=1)
    params, table = co.params(c), co.table(yq, w)
    stream = co.host_stream(params, table, w)
    p = MicroCellPlan(dim, n, "poisson")
    M = np.eye(dim)[None] + 0.3 * rng.standard_normal((nc, dim, dim))
    for MM in (None, M):
        A, info = p.solve_separable(family, table, w, params, MM, return_info=True)
        assert not info.any()
        assert np.array_equal(A, p.solve(stream, MM))
    # and the stream is what the generic callable path samples, to rounding
    generic = np.stack([np.tensordot(w, np.asarray(co(c[k], yq.reshape(-1, dim).T)).reshape(yq.shape[:2]), axes=([0], [1])) for k in range(nc)])
    assert np.abs(stream - generic).max() < 1e-14 * np.abs(generic).max()
    with pytest.raises(ValueError):
        p.solve_separable(family, table[:-1], w, params)


def test_separable_in_poisson_hmm_matches_callable_and_analytic():
    """Analytic example 2 of the reference (test_integration_poisson.py:146-185): A = 0.33 + 0.15 (sin 2 pi x0 + sin 2 pi y0)
    => A_H(x) = diag(sqrt(a^2 - 0.15^2), a), a = 0.33 + 0.15 sin 2 pi x0.  Device-sampled Separable == generic callable."""
    from hommx_amd import hmm, mesh

    msh, micro = mesh.create_unit_square(8, 8), mesh.create_unit_square(32, 32)
    sep = hmm.Separable("affine", lambda x: 0.33 + 0.15 * np.sin(2 * np.pi * x[0]), lambda x: 0.15, lambda y: np.sin(2 * np.pi * y[0]))
    gen = lambda x, y: 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))
    hs, hg = (hmm.PoissonHMM(msh, A, lambda x: 1.0, micro, 1.0 / 64) for A in (sep, gen))
    us, ug = hs.solve().x.array, hg.solve().x.array
    assert hs.quadrature_degree_used == 3 and hg.quadrature_degree_used == 3
    assert np.abs(hs.effective_tensors - hg.effective_tensors).max() < 1e-13
    assert np.abs(us - ug).max() < 1e-12 * np.abs(ug).max()
    a = 0.33 + 0.15 * np.sin(2 * np.pi * msh.cell_midpoints()[:, 0])
    assert np.abs(hs.effective_tensors[:, 0, 0] - np.sqrt(a * a - 0.15**2)).max() < 5e-4   # P1 discretisation error at n = 32 (observed 3.3e-4)
    assert np.abs(hs.effective_tensors[:, 1, 1] - a).max() < 1e-12
    # reciprocal family: analytic example 1 (test_integration_poisson.py:121-143): A = 1 / (2 + cos 2 pi y0) => diag(1/2, 1/sqrt 3)
    rec = hmm.Separable("reciprocal", lambda x: 2.0, lambda x: 1.0, lambda y: np.cos(2 * np.pi * y[0]))
    hr = hmm.PoissonHMM(mesh.create_unit_square(2, 2), rec, lambda x: 1.0, micro, 1.0 / 64)
    hr.solve()
    assert np.abs(hr.effective_tensors[:, 0, 0] - 0.5).max() < 5e-4  # P1 discretisation error at n = 32 (observed 2.2e-4)
    assert np.abs(hr.effective_tensors[:, 1, 1] - 1.0 / np.sqrt(3.0)).max() < 1e-6


@pytest.mark.parametrize("dim,n", [(2, 10), (2, 32), (3, 6)])
def test_separable_isotropic_elasticity_equals_host_stream_bitwise(dim, n, rng):
    """(lambda, mu) = a(x) + b(x) g(y) per Lame parameter (the reference's 2D beam coefficient, test_integration_linear_elasticity.py:78-93):
    device sampler == plan.solve of the stream the host forms with the same formula, on the one-wave kernel (b = 20), the LDS kernel
    (b = 64) and the multifrontal route (3D, b = 108)."""
    from hommx_amd import MicroCellPlan, hmm

    yq, w = _tables(dim, n)
    co = hmm.Separable("affine", lambda x: hmm.Lame(1.25 + 0.5 * x[0], 5.0 + x[1]), lambda x: hmm.Lame(0.3 * x[1], 4.5 - x[0]),
                       lambda y: np.sin(2 * np.pi * y[0]) * np.cos(2 * np.pi * y[dim - 1]))
    nc = 6
    c = np.concatenate([rng.uniform(size=(nc, 2)), np.zeros((nc, 1))], axis=1)
    params, table = co.params(c), co.table(yq, w)
    assert params.shape == (nc, 2, 2)
    stream = co.host_stream(params, table, w)
    assert stream.shape == (nc, yq.shape[0], 2) and stream[..., 1].min() > 0
    p = MicroCellPlan(dim, n, "elasticity")
    M = np.eye(dim)[None] + 0.2 * rng.standard_normal((nc, dim, dim))
    for MM in (None, M):
        A, info = p.solve_separable("affine", table, w, params, MM, return_info=True)
        assert not info.any()
        assert np.array_equal(A, p.solve(stream, MM))
    # the stream is what the generic callable path samples, to rounding
    v = co(c[0], yq.reshape(-1, dim).T)
    mu = np.tensordot(w, np.asarray(v.mu).reshape(yq.shape[:2]), axes=([0], [1]))
    assert np.abs(stream[0, :, 1] - mu).max() < 1e-14 * np.abs(mu).max()
    with pytest.raises(Exception):
        p.solve_separable("reciprocal", np.ones((p.n_el, len(w))), w, params)   # Lame-valued coefficients are affine
