"""Synthetic inputs of the five BASELINE.json configurations.

Coefficient families come from the reference's examples: laminate (examples/diffusion/laminate.py:101-102),
wrapped-disc inclusion (examples/diffusion/inclusion.py:107-118), fibre Hooke tensor
(examples/linear_elasticity/rotated_fibers.py:23-38, 66-76); the theta maps from README.md:96-99, 171-185.
All of them are piecewise constant in y, so UFL's degree estimation gives the centroid rule and the
element mean is the value at the element barycentre.

Everything is generated with NumPy on the host (deterministic, no RNG); the arrays are what
``MicroCellPlan.solve`` takes.
"""

from __future__ import annotations

import numpy as np

from . import mesh as _mesh


def element_barycentres(dim: int, n: int) -> np.ndarray:
    """Barycentres y_K[n_el, dim] of the DOLFINx-style unit-cell mesh, element order n_sub*(cell)+s."""
    m = _mesh.create_unit_square(n, n) if dim == 2 else _mesh.create_unit_cube(n, n, n)
    return m.cell_midpoints()[:, :dim]


def wrapped_disc(u, v, r=0.25):
    """1-periodic disc of radius r about (1/2,1/2): inclusion.py:107-114, rotated_fibers.py:23-29."""
    du = np.arccos(np.cos(2 * np.pi * (u - 0.5)))
    dv = np.arccos(np.cos(2 * np.pi * (v - 0.5)))
    return du**2 + dv**2 < (2 * np.pi) ** 2 * r**2


def c1_laminate(nx=8, n=16):
    """C1: 2D PoissonHMM, 8x8 macro, 16x16 micro, laminate made x-dependent so that cells differ."""
    msh = _mesh.create_unit_square(nx, nx)
    c = msh.cell_midpoints()
    y = element_barycentres(2, n)
    hi = 5.0 * (1.0 + c[:, 0])
    lo = 0.05 * (1.0 + c[:, 1])
    coef = np.where(np.cos(2 * np.pi * y[None, :, 0]) < 0, hi[:, None], lo[:, None])
    return msh, coef, None


def c1_exact(msh):
    c = msh.cell_midpoints()
    hi = 5.0 * (1.0 + c[:, 0])
    lo = 0.05 * (1.0 + c[:, 1])
    out = np.zeros((c.shape[0], 2, 2))
    out[:, 0, 0] = 2.0 / (1.0 / hi + 1.0 / lo)
    out[:, 1, 1] = 0.5 * (hi + lo)
    return out


def c2_inclusion(nx=64, n=32, x_shift=0.0):
    """C2 (headline): 2D PoissonHMM, 64x64 macro, 32x32 micro, inclusion a_in(x)=0.001(1+9 x0), 0.1 outside."""
    msh = _mesh.create_unit_square(nx, nx)
    c = msh.cell_midpoints()
    y = element_barycentres(2, n)
    inside = wrapped_disc(y[:, 0], y[:, 1])
    a_in = 0.001 * (1.0 + 9.0 * (c[:, 0] + x_shift))
    coef = np.where(inside[None, :], a_in[:, None], 0.1)
    return msh, coef, None


def c2_inclusion_two_phase(nx=64, n=32, x_shift=0.0, cells=None):
    """C2 as (mesh, mask[n_el], values[N_c, 2]): what ``MicroCellPlan.solve_two_phase`` takes (phase 0 = matrix, 1 = disc).
    ``cells``: optional index array -- sample only these macro cells (a rank's shard)."""
    msh = _mesh.create_unit_square(nx, nx)
    c = msh.cell_midpoints()
    if cells is not None:
        c = c[cells]
    y = element_barycentres(2, n)
    mask = wrapped_disc(y[:, 0], y[:, 1])
    values = np.stack([np.full(c.shape[0], 0.1), 0.001 * (1.0 + 9.0 * (c[:, 0] + x_shift))], axis=1)
    return msh, mask, values


def c3_wavy_laminate(nx=128, n=32):
    """C3: PoissonStratifiedHMM, laminate in y1, theta(x) = (x0, x1 - sin 2 pi x0) (README.md:96-99)."""
    msh = _mesh.create_unit_square(nx, nx)
    c = msh.cell_midpoints()
    y = element_barycentres(2, n)
    coef = np.broadcast_to(np.where(np.cos(2 * np.pi * y[:, 1]) < 0, 5.0, 0.05), (c.shape[0], y.shape[0])).copy()
    M = np.zeros((c.shape[0], 2, 2))
    M[:, 0, 0] = 1.0
    M[:, 0, 1] = -2 * np.pi * np.cos(2 * np.pi * c[:, 0])  # d theta_1 / d x_0
    M[:, 1, 1] = 1.0
    return msh, coef, M


def stratified_laminate_exact(M, a_hi=5.0, a_lo=0.05, layer_dir=1):
    """Closed form for a laminate layered in direction k: A_H = <a>(I - m m^T/|m|^2) + a_harm m m^T/|m|^2,
    m = M e_k."""
    mean = 0.5 * (a_hi + a_lo)
    harm = 2.0 / (1.0 / a_hi + 1.0 / a_lo)
    m = M[:, :, layer_dir]
    P = m[:, :, None] * m[:, None, :] / np.sum(m * m, axis=1)[:, None, None]
    return mean * (np.eye(M.shape[1])[None] - P) + harm * P


def fibre_lame(c, n, mu_in):
    y = element_barycentres(3, n)
    inside = wrapped_disc(y[:, 1], y[:, 2])
    mu = np.where(inside[None, :], np.asarray(mu_in)[:, None], 0.001)
    lam = np.ones_like(mu)
    return np.stack([lam, mu], axis=-1)


def fibre_mask(n):
    """Phase mask of the fibre (wrapped disc in (y1, y2)) on the 6 n^3 tets of the unit-cell mesh."""
    y = element_barycentres(3, n)
    return wrapped_disc(y[:, 1], y[:, 2])


def c5_theta_transpose(c):
    """M[i][j] = d theta_j / d x_i at the points c[N, 3] for theta(x) = (x0, x1, cos(g) x2 - sin(g) x0), g = (pi/2) x1/0.4."""
    gam = 0.5 * np.pi * c[:, 1] / 0.4
    dg = 0.5 * np.pi / 0.4
    Dth = np.zeros((c.shape[0], 3, 3))  # Dth[i][j] = d theta_i / d x_j
    Dth[:, 0, 0] = 1.0
    Dth[:, 1, 1] = 1.0
    Dth[:, 2, 0] = -np.sin(gam)
    Dth[:, 2, 1] = dg * (-np.sin(gam) * c[:, 2] - np.cos(gam) * c[:, 0])
    Dth[:, 2, 2] = np.cos(gam)
    return np.transpose(Dth, (0, 2, 1)).copy()


def c4_two_phase(shape=(20, 6, 6), n=16, cells=None):
    """C4 as (mesh, mask[n_el], values[N_c, 2, 2] = (lambda, mu) of matrix / fibre, None)."""
    msh = _mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
    c = msh.cell_midpoints()
    if cells is not None:
        c = c[cells]
    values = np.empty((c.shape[0], 2, 2))
    values[:, :, 0] = 1.0
    values[:, 0, 1] = 0.001
    values[:, 1, 1] = 100.0 * (1.0 + c[:, 0])
    return msh, fibre_mask(n), values, None


def c5_two_phase(shape=(32, 16, 8), n=16, cells=None):
    """C5 as (mesh, mask[n_el], values[N_c, 2, 2], M[N_c, 3, 3]); ``cells`` restricts the sampling to a rank's shard."""
    msh = _mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
    c = msh.cell_midpoints()
    if cells is not None:
        c = c[cells]
    values = np.empty((c.shape[0], 2, 2))
    values[:, :, 0] = 1.0
    values[:, 0, 1] = 0.001
    values[:, 1, 1] = 100.0
    return msh, fibre_mask(n), values, c5_theta_transpose(c)


def c4_fibre_beam(shape=(20, 6, 6), n=16):
    """C4: LinearElasticityHMM, beam 1x0.4x0.1, fibre along y0 with mu_in = 100 (1 + x0)."""
    msh = _mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
    c = msh.cell_midpoints()
    return msh, fibre_lame(c, n, 100.0 * (1.0 + c[:, 0])), None


def c5_rotated_fibres(shape=(32, 16, 8), n=16):
    """C5: LinearElasticityStratifiedHMM, theta(x) = (x0, x1, cos(g) x2 - sin(g) x0), g = (pi/2) x1/0.4."""
    msh = _mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
    c = msh.cell_midpoints()
    coef = fibre_lame(c, n, np.full(c.shape[0], 100.0))
    return msh, coef, c5_theta_transpose(c)  # M[i][j] = d theta_j / d x_i (hmm.py:741)
