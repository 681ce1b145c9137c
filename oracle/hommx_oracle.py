"""CPU oracle for the HOMMX micro-cell hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy (float64) restatement of the per-macro-cell periodic micro
problem of flxrcz/hommx.  It is the *checker* for the HIP kernels, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``hommx_amd/`` imports or falls back to it.

Parity status
-------------
The reference (pure Python on DOLFINx / dolfinx_mpc / UFL / PETSc) cannot be imported in
this container (ordinary ModuleNotFoundError: none of those packages is installed and
there is no network -- SURVEY.md section 8(c)); it ships no golden vectors.  The oracle is
therefore pinned by the *known answers* the reference's own tests hold:

* test/integration/test_integration_poisson.py:121-143  (A_H = diag(1/2, 1/sqrt 3))
* test/integration/test_integration_poisson.py:146-185  (A_H(x) = diag(sqrt(a^2-.15^2), a))
* test/integration/test_integration_poisson.py:188-240  (S_loc == vol(T) G A_hom G^T)
* test/integration/test_integration_linear_elasticity.py:205-322 (C_H == C for constant C)
* test/unit/test_unit.py:25-103                         (periodic master/slave topology)

plus closed-form laminate results (see tests/test_oracle_kat.py).  Anything beyond those
(inclusion / fibre coefficients, the exact DOLFINx triangulation and quadrature point
sets, which are recalled from DOLFINx 0.9 / Basix 0.9 public sources and marked
"3P-memory" below) is **parity unpinned** by the reference and pinned only by this file.

A second restatement, independent in code and in algorithm (closed-form periodic stencil, dense block-cyclic elimination, Schur
form), lives in ``oracle/hommx_oracle_c.c`` for the 2D Poisson path; ``tests/test_oracle_c.py`` holds the two against each other.

Every function cites the reference lines (``/root/reference/src/hommx/...``) it restates.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------------------
# unit-cell mesh  (user code: mesh.create_unit_square / create_unit_cube on COMM_SELF,
# e.g. test_integration_poisson.py:81-83, 111-113; rotated_fibers.py:89)
# --------------------------------------------------------------------------------------


def unit_cell_mesh(dim: int, n: int):
    """P1 simplicial mesh of the unit cell Y=[0,1]^dim with n cells per side.

    3P-memory (DOLFINx 0.9 ``mesh::create_rectangle`` / ``create_box`` with the default
    ``DiagonalType.right``): vertex (i,j[,k]) has index i + (n+1) j [+ (n+1)^2 k]; each square
    (v0,v1,v2,v3) = (ll, lr, ul, ur) is cut into (v0,v1,v3), (v0,v2,v3); each cube into the six
    tetrahedra sharing the diagonal v0-v7 listed below.

    Returns (x[(n+1)^dim, dim], cells[n_el, dim+1]) with the element ordering used by the whole
    repository: element index = n_sub * (i + n j [+ n^2 k]) + t, n_sub = 2 (tri) / 6 (tet).
    """
    if dim == 2:
        ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="xy")
        x = np.stack([ii.ravel(), jj.ravel()], axis=1) / n
        ci, cj = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
        v0 = (cj * (n + 1) + ci).ravel()
        v1, v2 = v0 + 1, v0 + (n + 1)
        v3 = v1 + (n + 1)
        cells = np.stack(
            [np.stack([v0, v1, v3], axis=1), np.stack([v0, v2, v3], axis=1)], axis=1
        ).reshape(-1, 3)
        return x, cells
    if dim == 3:
        kk, jj, ii = np.meshgrid(np.arange(n + 1), np.arange(n + 1), np.arange(n + 1), indexing="ij")
        x = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1) / n
        ck, cj, ci = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        v0 = (ck * (n + 1) ** 2 + cj * (n + 1) + ci).ravel()
        v1, v2 = v0 + 1, v0 + (n + 1)
        v3 = v1 + (n + 1)
        v4, v5, v6, v7 = (v + (n + 1) ** 2 for v in (v0, v1, v2, v3))
        tets = [
            (v0, v1, v3, v7),
            (v0, v1, v7, v5),
            (v0, v5, v7, v4),
            (v0, v3, v2, v7),
            (v0, v6, v4, v7),
            (v0, v2, v6, v7),
        ]
        cells = np.stack([np.stack(t, axis=1) for t in tets], axis=1).reshape(-1, 4)
        return x, cells
    raise ValueError("dim must be 2 or 3")


def periodic_master_map(dim: int, n: int) -> np.ndarray:
    """Node -> periodic unknown index (cell_problem.py:38-136 in 2D, :139-300 in 3D).

    The reference constrains every dof on a max-face to its image on the min-face (faces first,
    then doubly constrained edges, then the triply constrained corner, so that no slave is also a
    master).  On the structured unit cell the net effect is the torus identification
    (i,j[,k]) -> (i mod n, j mod n[, k mod n]); the (n+1)^dim nodes collapse to n^dim unknowns.
    """
    idx = np.arange(n + 1) % n
    if dim == 2:
        return (idx[None, :] + n * idx[:, None]).ravel()
    return (idx[None, None, :] + n * idx[None, :, None] + n * n * idx[:, None, None]).ravel()


def periodic_slaves_masters(dim: int, n: int):
    """(slave node, master node) pairs exactly as cell_problem.py builds them, in node numbering
    of :func:`unit_cell_mesh`.  Used to check the topology pins of test/unit/test_unit.py:25-103."""
    x, _ = unit_cell_mesh(dim, n)
    g = np.rint(x * n).astype(int)
    on_max = g == n
    slaves = np.nonzero(on_max.any(axis=1))[0]
    gm = g[slaves].copy()
    gm[on_max[slaves]] = 0
    stride = (n + 1) ** np.arange(dim)
    masters = gm @ stride
    return slaves, masters


# --------------------------------------------------------------------------------------
# quadrature (what UFL/FFCx/Basix pick for the coefficient A(c_T, y))
# --------------------------------------------------------------------------------------


def quadrature_rule(dim: int, degree: int):
    """Reference-simplex rule (barycentric points[nq, dim+1], weights[nq] summing to 1).

    3P-memory: Basix default rules.  degree<=1: centroid.  degree 2: 3 / 4 interior points.
    Triangle degree 3: 6-point Strang-Fix rule.  Tetrahedron degree 3: Basix uses a 6-point
    Xiao-Gimbutas rule whose digits are not available offline, so degree 3 in 3D uses the 8-point
    collapsed Gauss-Jacobi rule (also exact to degree 3, positive weights -- the classical 5-point
    Keast rule has a weight of -0.8) -- parity unpinned for that case.

    Only the element mean of A enters the discrete problem (P1 gradients are element-wise
    constant), so the rule only matters for non-piecewise-constant A.
    """
    if degree <= 1:
        return np.full((1, dim + 1), 1.0 / (dim + 1)), np.ones(1)
    if dim == 2 and degree == 2:
        p = np.full((3, 3), 1.0 / 6.0)
        np.fill_diagonal(p, 2.0 / 3.0)
        return p, np.full(3, 1.0 / 3.0)
    if dim == 2:
        a, b, c = 0.659027622374092, 0.231933368553031, 0.109039009072877
        p = np.array(list(itertools.permutations((a, b, c))))
        return p, np.full(6, 1.0 / 6.0)
    if dim == 3 and degree == 2:
        a, b = 0.5854101966249685, 0.1381966011250105
        p = np.full((4, 4), b)
        np.fill_diagonal(p, a)
        return p, np.full(4, 0.25)
    return _collapsed_gauss_jacobi_tet(2)


def _collapsed_gauss_jacobi_tet(m: int):
    """m^3-point collapsed Gauss-Jacobi rule on the tetrahedron (exact to degree 2m - 1, positive weights): barycentric points
    and weights summing to 1."""
    from scipy.special import roots_jacobi, roots_legendre

    x0, w0 = roots_jacobi(m, 2.0, 0.0)
    x1, w1 = roots_jacobi(m, 1.0, 0.0)
    x2, w2 = roots_legendre(m)
    a = 0.5 * (x0 + 1.0)[:, None, None] * np.ones((1, m, m))
    b = 0.5 * (x1 + 1.0)[None, :, None] * np.ones((m, 1, m))
    c = 0.5 * (x2 + 1.0)[None, None, :] * np.ones((m, m, 1))
    l1, l2, l3 = a, (1.0 - a) * b, (1.0 - a) * (1.0 - b) * c
    w = (w0[:, None, None] * w1[None, :, None] * w2[None, None, :]).ravel()
    return np.stack([1.0 - l1 - l2 - l3, l1, l2, l3], axis=-1).reshape(-1, 4), w / w.sum()


def element_quadrature_points(dim: int, n: int, degree: int):
    """Physical quadrature points y_q[n_el, nq, dim] on the unit-cell mesh + weights[nq]."""
    x, cells = unit_cell_mesh(dim, n)
    bary, w = quadrature_rule(dim, degree)
    yq = np.einsum("qa,eak->eqk", bary, x[cells])
    return yq, w


def sample_coefficient(A, c_T: np.ndarray, dim: int, n: int, degree: int = 0) -> np.ndarray:
    """Element means  Abar_K = sum_q w_q A(c_T, y_q)  of the coefficient placeholder
    ``self._A_micro = A(x_macro, y)`` (hmm.py:190-198) with ``x_macro.value = c_T`` (hmm.py:349-352).

    ``A(x, y)`` is called NumPy-vectorised: x has shape (dim,), y has shape (dim, npts) and the
    result has shape (npts,) + tensor shape (or is a scalar, which is broadcast).
    Returns coef[n_el, ...].
    """
    yq, w = element_quadrature_points(dim, n, degree)
    n_el, nq, _ = yq.shape
    vals = np.asarray(A(np.asarray(c_T, dtype=float), yq.reshape(-1, dim).T), dtype=float)
    if vals.ndim == 0 or vals.shape[0] != n_el * nq:
        vals = np.broadcast_to(vals, (n_el * nq,) + vals.shape).copy()
    vals = vals.reshape((n_el, nq) + vals.shape[1:])
    return np.tensordot(w, vals, axes=([0], [1]))


# --------------------------------------------------------------------------------------
# tensors
# --------------------------------------------------------------------------------------


def voigt_pairs(dim: int):
    """Index pairs of the t = dim(dim+1)/2 canonical symmetric unit strains E^m
    (order 00, 11, [22,] then the off-diagonals 01[, 02, 12])."""
    if dim == 2:
        return [(0, 0), (1, 1), (0, 1)]
    return [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]


def unit_strains(dim: int) -> np.ndarray:
    """E[m] = sym(e_k (x) e_l) for the Voigt pair m = (k,l)   (tensorial, not engineering)."""
    pairs = voigt_pairs(dim)
    E = np.zeros((len(pairs), dim, dim))
    for m, (k, l) in enumerate(pairs):
        E[m, k, l] += 0.5
        E[m, l, k] += 0.5
    return E


def isotropic_hooke(lam, mu, dim: int) -> np.ndarray:
    """A_ijkl = lam d_ij d_kl + mu (d_ik d_jl + d_il d_jk)
    (test_integration_linear_elasticity.py:84-93, 227-236; rotated_fibers.py:66-76)."""
    lam = np.asarray(lam, dtype=float)
    mu = np.asarray(mu, dtype=float)
    I = np.eye(dim)
    t1 = np.einsum("ij,kl->ijkl", I, I)
    t2 = np.einsum("ik,jl->ijkl", I, I) + np.einsum("il,jk->ijkl", I, I)
    return lam[..., None, None, None, None] * t1 + mu[..., None, None, None, None] * t2


# --------------------------------------------------------------------------------------
# the micro problem
# --------------------------------------------------------------------------------------


@dataclass
class CellProblem:
    """Discrete periodic micro problem of ONE macro cell (hmm.py:334-369, cell_problem.py:303-388)."""

    dim: int
    n: int
    bs: int  # 1 Poisson, dim elasticity
    K: sp.csr_matrix  # [n_dof, n_dof] periodic stiffness (hmm.py:644-647 / 759-766 / 891-896 / 1032-1041)
    B: np.ndarray  # [n_dof, t] canonical load vectors (hmm.py:649-650 / 768-772 / 898-903 / 1043-1048)
    C0: np.ndarray  # [t, t]  int_Y  E^m : A : E^n  (the corrector-free part of hmm.py:652-667 etc.)
    xi: np.ndarray  # [n_el, dim+1, bs, ...] generalised element "gradients" (for the energy form)
    coefT: np.ndarray  # [n_el, ...] element tensor
    vol: np.ndarray  # [n_el]
    cells_p: np.ndarray  # [n_el, dim+1] periodic node ids


def _element_geometry(dim: int, n: int):
    x, cells = unit_cell_mesh(dim, n)
    X = x[cells]  # [n_el, dim+1, dim]
    # P1 gradients: rows of inverse of [1 | X]
    ones = np.ones(X.shape[:2] + (1,))
    Minv = np.linalg.inv(np.concatenate([ones, X], axis=2))  # [n_el, dim+1(coef), dim+1(node)]
    grads = np.transpose(Minv[:, 1:, :], (0, 2, 1))  # [n_el, node a, dim]
    J = X[:, 1:, :] - X[:, :1, :]
    fact = 2.0 if dim == 2 else 6.0
    vol = np.abs(np.linalg.det(J)) / fact
    cells_p = periodic_master_map(dim, n)[cells]
    return grads, vol, cells_p


def build_cell_problem(kind: str, dim: int, n: int, coef: np.ndarray, M: np.ndarray | None = None) -> CellProblem:
    """Assemble the periodic micro problem for one macro cell.

    kind:
      'poisson'     coef[n_el] scalar or coef[n_el, dim, dim] matrix-valued A      (hmm.py:644-667;
                    with M != None the stratified forms hmm.py:759-789)
      'elasticity'  coef[n_el, 2] = (lambda, mu) isotropic or coef[n_el, dim,dim,dim,dim] (hmm.py:887-922;
                    with M != None hmm.py:1024-1067: e_D(u) = sym(M . nabla_grad u))
    M = Dtheta_transpose(c_T), M[i, j] = d theta_j / d x_i (hmm.py:741), or None for identity.

    Periodic identification = ``assemble_matrix(a, mpc)`` / ``assemble_vector(L, mpc)`` of
    cell_problem.py:367-375 on the torus map of :func:`periodic_master_map`.
    """
    grads, vol, cells_p = _element_geometry(dim, n)
    n_el = grads.shape[0]
    if M is None:
        M = np.eye(dim)
    M = np.asarray(M, dtype=float)
    gt = np.einsum("ik,eak->eai", M, grads)  # g~_a = M g_a
    coef = np.asarray(coef, dtype=float)
    nn = n**dim
    if kind == "poisson":
        bs = 1
        A = coef if coef.ndim == 3 else coef[:, None, None] * np.eye(dim)
        # K_ab = vol g~_a . A g~_b ; b_{a,m} = - vol (A e_m) . g~_a ; C0 = sum vol A
        Ke = np.einsum("e,eai,eij,ebj->eab", vol, gt, A, gt)
        Be = -np.einsum("e,eai,eim->eam", vol, gt, A)
        C0 = np.einsum("e,eij->ij", vol, A)
        xi = gt[:, :, None, :]  # [e, a, 1, dim]
        coefT = A
        rows = cells_p
    elif kind == "elasticity":
        bs = dim
        C = isotropic_hooke(coef[:, 0], coef[:, 1], dim) if coef.ndim == 2 else coef
        E = unit_strains(dim)
        # eps_{a alpha} = sym(e_alpha (x) g~_a)
        I = np.eye(dim)
        eps = 0.5 * (np.einsum("pi,eaj->eapij", I, gt) + np.einsum("pj,eai->eapij", I, gt))
        Ke = np.einsum("e,eapij,eijkl,ebqkl->eapbq", vol, eps, C, eps)
        Be = -np.einsum("e,eapij,eijkl,mkl->eapm", vol, eps, C, E)
        C0 = np.einsum("e,mij,eijkl,nkl->mn", vol, E, C, E)
        Ke = Ke.reshape(n_el, (dim + 1) * bs, (dim + 1) * bs)
        Be = Be.reshape(n_el, (dim + 1) * bs, -1)
        xi = eps  # [e, a, alpha, i, j]
        coefT = C
        rows = (cells_p[:, :, None] * bs + np.arange(bs)[None, None, :]).reshape(n_el, -1)
    else:
        raise ValueError(kind)
    n_dof = nn * bs
    nl = rows.shape[1]
    r = np.repeat(rows, nl, axis=1).ravel()
    c = np.tile(rows, (1, nl)).ravel()
    K = sp.coo_matrix((Ke.ravel(), (r, c)), shape=(n_dof, n_dof)).tocsr()
    t = Be.shape[2]
    B = np.zeros((n_dof, t))
    for m in range(t):
        B[:, m] = np.bincount(rows.ravel(), weights=Be[:, :, m].ravel(), minlength=n_dof)
    return CellProblem(dim, n, bs, K, B, C0, xi, coefT, vol, cells_p)


def solve_correctors(cp: CellProblem, rhs: np.ndarray | None = None) -> np.ndarray:
    """Solve K chi = b modulo the kernel (cell_problem.py:345-388).

    The reference attaches the constant vector as (near-)nullspace and runs a Krylov method
    (default GMRES+ILU rtol 1e-5, hmm.py:153-155) or LU (test_integration_poisson.py:207-211).
    Here: direct sparse LU of the system with node 0 pinned (all bs components); every gauge
    gives the same effective tensor because only strains of chi enter.
    """
    b = cp.B if rhs is None else rhs
    n_dof = cp.K.shape[0]
    if n_dof == cp.bs:  # n == 1: a single periodic node, K == 0
        return np.zeros_like(b)
    keep = np.arange(cp.bs, n_dof)
    Kr = cp.K[keep][:, keep].tocsc()
    lu = spla.splu(Kr)
    chi = np.zeros_like(b)
    chi[keep] = lu.solve(np.ascontiguousarray(b[keep]))
    return chi


def effective_tensor(cp: CellProblem, chi: np.ndarray | None = None, form: str = "energy") -> np.ndarray:
    """A_H / C_H [t, t] of one cell.

    form='energy':  (E^m + eps(chi_m)) : A : (E^n + eps(chi_n)) integrated element by element
                    -- the literal functional of hmm.py:652-667 / 774-789 / 905-922 / 1050-1067
                    divided by eps^2 and vol(Y)=1 (hmm.py:101, 366-369).
    form='schur':   C0 - B^T K^+ B  (algebraically identical; what the HIP kernels evaluate).
    """
    if chi is None:
        chi = solve_correctors(cp)
    if form == "schur":
        return cp.C0 - cp.B.T @ chi
    dim, bs = cp.dim, cp.bs
    t = cp.B.shape[1]
    if bs == 1:
        # flux-like field per element: F[e, m, :] = e_m + sum_a chi_m[a] g~_a
        ce = chi[cp.cells_p]  # [e, a, t]
        F = np.einsum("eam,eai->emi", ce, cp.xi[:, :, 0, :]) + np.eye(dim)[None, :t, :]
        return np.einsum("e,emi,eij,enj->mn", cp.vol, F, cp.coefT, F)
    E = unit_strains(dim)
    ce = chi.reshape(-1, bs, t)[cp.cells_p]  # [e, a, alpha, t]
    F = np.einsum("eapm,eapij->emij", ce, cp.xi) + E[None]
    return np.einsum("e,emij,eijkl,enkl->mn", cp.vol, F, cp.coefT, F)


def effective_tensor_batch(kind, dim, n, coef, M=None, form="energy") -> np.ndarray:
    """Loop of :func:`effective_tensor` over macro cells: coef[N_c, n_el, ...], M[N_c, dim, dim] or None."""
    out = []
    for c in range(coef.shape[0]):
        cp = build_cell_problem(kind, dim, n, coef[c], None if M is None else M[c])
        out.append(effective_tensor(cp, form=form))
    return np.stack(out)


# --------------------------------------------------------------------------------------
# macro-cell side of the hot path (hmm.py:20-28, 334-369)
# --------------------------------------------------------------------------------------


def simplex_volume(X: np.ndarray) -> float:
    """hmm.py:20-28 (_triangle_area / _tetrahedron_volume)."""
    d = X.shape[1]
    J = X[1:] - X[0]
    return abs(np.linalg.det(J)) / (2.0 if d == 2 else 6.0)


def p1_gradients(X: np.ndarray) -> np.ndarray:
    """G[a, :] = grad phi_a of the P1 basis on the simplex with vertices X[a]."""
    d = X.shape[1]
    Minv = np.linalg.inv(np.concatenate([np.ones((d + 1, 1)), X], axis=1))
    return Minv[1:, :].T


def local_stiffness_from_tensor(kind: str, X: np.ndarray, AH: np.ndarray) -> np.ndarray:
    """S_loc = vol(T) G A_H G^T (Poisson) or vol(T) sym(e_al (x) grad phi_a):C_H:sym(e_be (x) grad phi_b)
    with local dof index a*bs+alpha (hmm.py:31-40, 361-369)."""
    d = X.shape[1]
    G = p1_gradients(X)
    vol = simplex_volume(X)
    if kind == "poisson":
        return vol * G @ AH @ G.T
    pairs = voigt_pairs(d)
    # strain of basis (a, alpha) expressed in the unit-strain basis: e = sum_m w_m E^m, with
    # coefficient e_kl for diagonal pairs and 2 e_kl for off-diagonal pairs.
    I = np.eye(d)
    eps = 0.5 * (np.einsum("pi,aj->apij", I, G) + np.einsum("pj,ai->apij", I, G)).reshape(-1, d, d)
    W = np.stack([eps[:, k, l] * (1.0 if k == l else 2.0) for (k, l) in pairs], axis=1)
    return vol * W @ AH @ W.T


def local_stiffness_reference_shaped(
    kind: str, dim: int, n: int, X: np.ndarray, coef: np.ndarray, eps: float, M: np.ndarray | None = None
) -> np.ndarray:
    """Step-for-step restatement of ``BaseHMM._compute_local_stiffness`` (hmm.py:334-369):

    for every macro basis function i (nb = (dim+1)*bs of them, hmm.py:354):
       v_micro_i = P1 interpolant on the micro mesh of phi_i(c_T + eps (y - ybar))   (hmm.py:371-395)
       solve  a(chi_i, z) = -l(v_micro_i; z)                                          (hmm.py:397-432)
    then  S_loc[i, j] = eps^-2 int_Y A (D v_i + D~ chi_i) : (D v_j + D~ chi_j)        (hmm.py:361-364)
    scaled by vol(T)/vol(Y)                                                           (hmm.py:366-369).

    This is the slow, literal form (nb solves, nb^2 energies); it exists to prove that the compact
    form used everywhere else (d or d(d+1)/2 canonical solves + S_loc = vol(T) G A_H G^T) is the same
    numbers.
    """
    x, cells = unit_cell_mesh(dim, n)
    grads, vol, cells_p = _element_geometry(dim, n)
    cp = build_cell_problem(kind, dim, n, coef, M)
    bs = cp.bs
    nb = (dim + 1) * bs
    c_T = X.mean(axis=0)
    ybar = x.mean(axis=0)
    pts = (x - ybar) * eps + c_T  # hmm.py:392
    # macro P1 basis evaluated through the affine pull-back of THIS cell (Function.eval with cells=cell_index)
    Minv = np.linalg.inv(np.concatenate([np.ones((dim + 1, 1)), X], axis=1))
    phi = np.concatenate([np.ones((pts.shape[0], 1)), pts], axis=1) @ Minv  # [node, a]
    Mm = np.eye(dim) if M is None else np.asarray(M, float)
    # plain (un-stratified) element gradients of the interpolated macro function
    Dv = []  # per basis function: element-wise "macro strain" tensor
    rhs = np.zeros((cp.K.shape[0], nb))
    n_el = cells.shape[0]
    if kind == "poisson":
        A = cp.coefT
        for a in range(dim + 1):
            gv = np.einsum("ea,eai->ei", phi[cells, a], grads)  # grad of P1 interpolant, [e, dim]
            Dv.append(gv)
            be = -np.einsum("e,ei,eij,eaj->ea", vol, gv, A, cp.xi[:, :, 0, :])
            rhs[:, a] = np.bincount(cells_p.ravel(), weights=be.ravel(), minlength=cp.K.shape[0])
        chi = solve_correctors(cp, rhs)
        F = [Dv[i] + np.einsum("ea,eai->ei", chi[cells_p, i], cp.xi[:, :, 0, :]) for i in range(nb)]
        S = np.array([[np.einsum("e,ei,eij,ej->", vol, F[i], A, F[j]) for j in range(nb)] for i in range(nb)])
    else:
        C = cp.coefT
        rows = (cells_p[:, :, None] * bs + np.arange(bs)[None, None, :]).reshape(n_el, -1)
        for a in range(dim + 1):
            for al in range(bs):
                gv = np.einsum("ea,eai->ei", phi[cells, a], grads)
                g = np.zeros((n_el, dim, dim))
                g[:, al, :] = gv  # grad(u)[i,j] = d u_i / d x_j  (ufl.grad), u = phi_a e_al
                ev = 0.5 * (g + np.transpose(g, (0, 2, 1)))  # _e(v_micro), hmm.py:887-889 (plain, also stratified: :1046)
                Dv.append(ev)
                be = -np.einsum("e,ekl,eijkl,eapij->eap", vol, ev, C, cp.xi).reshape(n_el, -1)
                rhs[:, a * bs + al] = np.bincount(rows.ravel(), weights=be.ravel(), minlength=cp.K.shape[0])
        chi = solve_correctors(cp, rhs)
        F = []
        for i in range(nb):
            ce = chi[:, i].reshape(-1, bs)[cells_p]  # [e, a, alpha]
            F.append(Dv[i] + np.einsum("eap,eapij->eij", ce, cp.xi))
        S = np.array([[np.einsum("e,ekl,eijkl,eij->", vol, F[i], C, F[j]) for j in range(nb)] for i in range(nb)])
    S /= eps**2
    return S * simplex_volume(X) / vol.sum()
