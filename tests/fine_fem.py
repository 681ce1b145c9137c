"""Fine-scale P1 reference solver for the heterogeneous problem itself -- test infrastructure.

The reference's integration tests (test_integration_poisson.py:322-572) accept an HMM solution when it is close, in relative L2 norm, to a
plain P1 finite-element solution of  -div(A(x, x / eps) grad u) = f  on a mesh that resolves eps (1024 x 1024 there, DOLFINx + PETSc CG / GAMG).
This module is that fine-scale solve with NumPy / SciPy only: the same triangulation (squares split along the right diagonal), the element
means of the coefficient by a degree-4 six-point rule (only the mean enters: P1 gradients are element-wise constant), conjugate gradients
preconditioned by a geometric multigrid V-cycle (bilinear prolongation, Galerkin coarse operators, damped Jacobi).  It shares no code with
hommx_amd or with the oracle: an independent discretisation of the ORIGINAL problem, against which the whole HMM pipeline is judged."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

# six-point, degree-4 triangle rule (Dunavant), barycentric points and weights
_A1, _B1, _W1 = 0.445948490915965, 0.108103018168070, 0.223381589678011
_A2, _B2, _W2 = 0.091576213509771, 0.816847572980459, 0.109951743655322
_BARY = np.array([[_A1, _A1, _B1], [_A1, _B1, _A1], [_B1, _A1, _A1], [_A2, _A2, _B2], [_A2, _B2, _A2], [_B2, _A2, _A2]])
_WQ = np.array([_W1, _W1, _W1, _W2, _W2, _W2])


def _interp_1d(nf: int) -> sp.csr_matrix:
    """Linear interpolation from the nf/2 - 1 interior nodes of the coarse 1D grid to the nf - 1 interior nodes of the fine one."""
    nc = nf // 2
    rows, cols, vals = [], [], []
    for i in range(1, nf):          # fine interior node i
        if i % 2 == 0:
            rows.append(i - 1); cols.append(i // 2 - 1); vals.append(1.0)
        else:
            for c in ((i - 1) // 2, (i + 1) // 2):
                if 1 <= c <= nc - 1:
                    rows.append(i - 1); cols.append(c - 1); vals.append(0.5)
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf - 1, nc - 1))


class _Multigrid:
    def __init__(self, K: sp.csr_matrix, n: int, coarsest: int = 16, dim: int = 2):
        self.levels = []
        while True:
            D = K.diagonal()
            if n <= coarsest:
                self.levels.append((K, D, None))
                self.coarse = spl.splu(K.tocsc())
                break
            I1 = _interp_1d(n)
            P = sp.kron(I1, I1, format="csr")
            if dim == 3:
                P = sp.kron(P, I1, format="csr")
            self.levels.append((K, D, P))
            K = (P.T @ K @ P).tocsr()
            n //= 2

    def vcycle(self, b: np.ndarray, lvl: int = 0) -> np.ndarray:
        K, D, P = self.levels[lvl]
        if P is None:
            return self.coarse.solve(b)
        w = 0.7
        x = w * b / D
        x += w * (b - K @ x) / D
        xc = self.vcycle(P.T @ (b - K @ x), lvl + 1)
        x += P @ xc
        x += w * (b - K @ x) / D
        x += w * (b - K @ x) / D
        return x


def solve_fine(N: int, A_of_x, f_val: float, g_of_x, rtol: float = 1e-10):
    """u[(N+1), (N+1)] at the nodes (i / N, j / N) of the P1 solution of -div(A grad u) = f_val on the unit square, u = g on the boundary.
    N must be a power of two; A_of_x and g_of_x take an array [2, ...].

    On squares split along the right diagonal the P1 stiffness matrix is a five-point stencil: with aL / aU the coefficient means of the
    lower triangle (v00, v10, v11) / upper triangle (v00, v01, v11) of a square, the edge v00-v10 carries aL / 2, v01-v11 aU / 2,
    v00-v01 aU / 2, v10-v11 aL / 2 and the diagonal edge nothing (its two basis gradients are orthogonal); h cancels."""
    h = 1.0 / N
    xc, yc = np.meshgrid(np.arange(N) * h, np.arange(N) * h, indexing="ij")  # lower-left corners of the squares
    aL = np.zeros((N, N))
    aU = np.zeros((N, N))
    for q in range(6):
        l0, l1, l2 = _BARY[q]
        aL += _WQ[q] * A_of_x(np.stack([xc + h * (l1 + l2), yc + h * l2]))
        aU += _WQ[q] * A_of_x(np.stack([xc + h * l2, yc + h * (l1 + l2)]))
    cH = np.zeros((N, N + 1))          # edge (i, j) - (i + 1, j)
    cH[:, :N] += 0.5 * aL
    cH[:, 1:] += 0.5 * aU
    cV = np.zeros((N + 1, N))          # edge (i, j) - (i, j + 1)
    cV[:N, :] += 0.5 * aU
    cV[1:, :] += 0.5 * aL
    u = np.zeros((N + 1, N + 1))
    xn, yn = np.meshgrid(np.arange(N + 1) * h, np.arange(N + 1) * h, indexing="ij")
    gb = g_of_x(np.stack([xn, yn]))
    u[0, :], u[N, :], u[:, 0], u[:, N] = gb[0, :], gb[N, :], gb[:, 0], gb[:, N]
    m = N - 1                          # interior nodes (i, j), 1 <= i, j <= N - 1, index (i - 1) m + (j - 1)
    cW, cE = cH[0:N - 1, 1:N], cH[1:N, 1:N]      # conductances to the west / east neighbour of interior node (i, j)
    cS, cNn = cV[1:N, 0:N - 1], cV[1:N, 1:N]     # south / north
    diag = (cW + cE + cS + cNn).ravel()
    east = -cE.copy(); east[-1, :] = 0.0         # (i, j) -> (i + 1, j): offset m
    north = -cNn.copy(); north[:, -1] = 0.0      # (i, j) -> (i, j + 1): offset 1
    e, nth = east.ravel()[: m * m - m], north.ravel()[: m * m - 1]
    Kff = sp.diags([diag, e, e, nth, nth], [0, m, -m, 1, -1], format="csr")
    rhs = np.full((m, m), f_val * h * h)
    rhs[0, :] += cW[0, :] * u[0, 1:N]
    rhs[-1, :] += cE[-1, :] * u[N, 1:N]
    rhs[:, 0] += cS[:, 0] * u[1:N, 0]
    rhs[:, -1] += cNn[:, -1] * u[1:N, N]
    mg = _Multigrid(Kff, N)
    M = spl.LinearOperator(Kff.shape, matvec=mg.vcycle)
    its = [0]
    sol, info = spl.cg(Kff, rhs.ravel(), rtol=rtol, atol=0.0, M=M, maxiter=200, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    if info != 0:
        raise RuntimeError(f"fine-scale CG did not converge (info {info})")
    u[1:N, 1:N] = sol.reshape(m, m)
    return u, its[0]


def _interp_1d_full(nf: int) -> sp.csr_matrix:
    """Linear interpolation from the nf/2 + 1 nodes of the coarse 1D grid (end nodes included: natural boundary) to the nf + 1 fine ones."""
    rows, cols, vals = [], [], []
    for i in range(nf + 1):
        if i % 2 == 0:
            rows.append(i); cols.append(i // 2); vals.append(1.0)
        else:
            rows += [i, i]; cols += [(i - 1) // 2, (i + 1) // 2]; vals += [0.5, 0.5]
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf + 1, nf // 2 + 1))


def solve_fine_darcy(N: int, A_of_x, f_val: float = 1.0, diagonal: str = "right", rtol: float = 1e-11):
    """The fine-scale reference solve of the reference's two diffusion examples (examples/diffusion/inclusion.py:140-161,
    laminate.py:131-144): P1 on create_unit_square(N, N), -div(A grad u) = f_val, "Darcy" data u = 1 on x0 = 0, u = 0 on x0 = 1
    (inclusion.py:64-88), natural conditions on x1 = 0, 1.  The coefficient is a ``ufl.conditional`` between constants, for which UFL
    estimates quadrature degree 0: ONE sample per triangle, at its centroid.  ``diagonal``: "right" (DOLFINx's default: squares split
    along v00 - v11) or "left" (v10 - v01), to see whether a reference-held number tells the two triangulations apart.
    Returns u[(N+1), (N+1)] at the nodes (i / N, j / N) and the CG iteration count."""
    h = 1.0 / N
    xc, yc = np.meshgrid(np.arange(N) * h, np.arange(N) * h, indexing="ij")  # lower-left corners of the squares
    third = h / 3.0
    cH = np.zeros((N, N + 1))          # edge (i, j) - (i + 1, j)
    cV = np.zeros((N + 1, N))          # edge (i, j) - (i, j + 1)
    if diagonal == "right":            # lower triangle (v00, v10, v11): legs bottom + right; upper (v00, v01, v11): legs left + top
        aL = A_of_x(np.stack([xc + 2 * third, yc + third]))
        aU = A_of_x(np.stack([xc + third, yc + 2 * third]))
        cH[:, :N] += 0.5 * aL; cH[:, 1:] += 0.5 * aU
        cV[:N, :] += 0.5 * aU; cV[1:, :] += 0.5 * aL
    elif diagonal == "left":           # (v00, v10, v01): legs bottom + left; (v10, v11, v01): legs right + top
        aL = A_of_x(np.stack([xc + third, yc + third]))
        aU = A_of_x(np.stack([xc + 2 * third, yc + 2 * third]))
        cH[:, :N] += 0.5 * aL; cH[:, 1:] += 0.5 * aU
        cV[:N, :] += 0.5 * aL; cV[1:, :] += 0.5 * aU
    else:
        raise ValueError(diagonal)
    mx, my = N - 1, N + 1              # unknowns (i, j), 1 <= i <= N - 1, 0 <= j <= N, index (i - 1) my + j
    cW, cE = cH[0:N - 1, :], cH[1:N, :]
    cS = np.zeros((mx, my)); cS[:, 1:] = cV[1:N, :]
    cNn = np.zeros((mx, my)); cNn[:, :N] = cV[1:N, :]
    diag = (cW + cE + cS + cNn).ravel()
    east = -cE.copy(); east[-1, :] = 0.0
    e = east.ravel()[: mx * my - my]
    nth = (-cNn).ravel()[: mx * my - 1]                       # zero where j = N: no coupling across rows of the index
    K = sp.diags([diag, e, e, nth, nth], [0, my, -my, 1, -1], format="csr")
    rhs = np.full((mx, my), f_val * h * h)
    rhs[:, 0] *= 0.5; rhs[:, -1] *= 0.5                       # three triangles instead of six around a node of the natural boundary
    rhs[0, :] += cW[0, :] * 1.0                               # u = 1 on x0 = 0; u = 0 on x0 = 1 adds nothing

    class _MG(_Multigrid):
        def __init__(self, K, n, coarsest=16):
            self.levels = []
            while True:
                D = K.diagonal()
                if n <= coarsest:
                    self.levels.append((K, D, None))
                    self.coarse = spl.splu(K.tocsc())
                    break
                P = sp.kron(_interp_1d(n), _interp_1d_full(n), format="csr")
                self.levels.append((K, D, P))
                K = (P.T @ K @ P).tocsr()
                n //= 2

    mg = _MG(K, N)
    M = spl.LinearOperator(K.shape, matvec=mg.vcycle)
    its = [0]
    sol, info = spl.cg(K, rhs.ravel(), rtol=rtol, atol=0.0, M=M, maxiter=400, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    if info != 0:
        raise RuntimeError(f"fine-scale CG did not converge (info {info})")
    u = np.zeros((N + 1, N + 1))
    u[0, :] = 1.0
    u[1:N, :] = sol.reshape(mx, my)
    return u, its[0]


# Kuhn (Freudenthal) split of a cube into six tetrahedra around the diagonal v000 - v111, one per order in which the path from v000 to v111
# takes its three axis steps (DOLFINx create_unit_cube).  The P1 gradients of such a tetrahedron are -e_p1, e_p1 - e_p2, e_p2 - e_p3, e_p3
# (over h): only consecutive path vertices couple, i.e. only axis-parallel edges carry a conductance, a_tet h / 6 per path step.
_PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
# four-point, degree-2 tetrahedron rule: barycentric (a, b, b, b) and permutations, weights 1/4
_TA, _TB = 0.5854101966249685, 0.1381966011250105


def solve_fine_3d(N: int, A_of_x, f_val: float, rtol: float = 1e-10):
    """u[(N+1)^3] at the nodes (i, j, k) / N of the P1 solution of -div(A grad u) = f_val on the unit cube with u = 0 on the boundary, on
    N^3 cubes of six Kuhn tetrahedra each.  N a power of two; A_of_x takes an array [3, ...]."""
    h = 1.0 / N
    c = np.stack(np.meshgrid(np.arange(N) * h, np.arange(N) * h, np.arange(N) * h, indexing="ij"))  # lower corners [3, N, N, N]
    # conductance of the edge from node p to p + e_ax, per axis: arrays over all nodes with a neighbour in that direction
    cond = [np.zeros((N + 1, N + 1, N + 1)) for _ in range(3)]
    E = np.eye(3)
    for perm in _PERMS:
        verts = [np.zeros(3)]
        for ax in perm:
            verts.append(verts[-1] + E[ax])
        verts = np.array(verts)                                  # 4 path vertices in units of h
        a = np.zeros((N, N, N))
        for q in range(4):
            lam = np.full(4, _TB)
            lam[q] = _TA
            pt = lam @ verts
            a += 0.25 * A_of_x(c + h * pt[:, None, None, None])
        for step, ax in enumerate(perm):                         # edge verts[step] -> verts[step + 1] along axis ax
            o = verts[step].astype(int)
            sl = tuple(slice(o[d], o[d] + N) for d in range(3))
            cond[ax][sl] += a * h / 6.0
    m = N - 1
    idx = lambda i, j, k: (i * m + j) * m + k                     # interior nodes 1..N-1 (shifted by one)
    diag = np.zeros((m, m, m))
    offs = []
    strides = (m * m, m, 1)
    for ax in range(3):
        cax = cond[ax]
        lo = [slice(1, N)] * 3
        hi = [slice(1, N)] * 3
        lo[ax] = slice(0, N - 1)                                 # edge towards the lower neighbour: starts at node - e_ax
        diag += cax[tuple(lo)] + cax[tuple(hi)]
        up = -cax[tuple(hi)].copy()                              # coupling node -> node + e_ax
        cut = [slice(None)] * 3
        cut[ax] = -1
        up[tuple(cut)] = 0.0
        offs.append((up.ravel()[: m**3 - strides[ax]], strides[ax]))
    K = sp.diags([diag.ravel()] + [o for o, _ in offs] + [o for o, _ in offs], [0] + [s for _, s in offs] + [-s for _, s in offs],
                 format="csr")
    rhs = np.full(m**3, f_val * h**3)                             # 24 tetrahedra of volume h^3 / 6 around a node, a quarter each
    mg = _Multigrid(K, N, coarsest=8, dim=3)
    M = spl.LinearOperator(K.shape, matvec=mg.vcycle)
    its = [0]
    sol, info = spl.cg(K, rhs, rtol=rtol, atol=0.0, M=M, maxiter=200, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    if info != 0:
        raise RuntimeError(f"fine-scale CG did not converge (info {info})")
    u = np.zeros((N + 1, N + 1, N + 1))
    u[1:N, 1:N, 1:N] = sol.reshape(m, m, m)
    return u, its[0]


def sample_p1_3d(u: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Values of the fine P1 function u[(N+1)^3] at pts[npoints, 3]: inside a cube, sort the local coordinates descending; the Kuhn
    tetrahedron is the path that takes the axis steps in that order."""
    N = u.shape[0] - 1
    s = np.clip(pts * N, 0.0, N)
    ijk = np.minimum(s.astype(int), N - 1)
    loc = s - ijk
    order = np.argsort(-loc, axis=1, kind="stable")
    val = u[ijk[:, 0], ijk[:, 1], ijk[:, 2]].copy()
    cur = ijk.copy()
    prev = val.copy()
    rows = np.arange(len(pts))
    for step in range(3):
        ax = order[:, step]
        cur[rows, ax] += 1
        nxt = u[cur[:, 0], cur[:, 1], cur[:, 2]]
        val += loc[rows, ax] * (nxt - prev)
        prev = nxt
    return val


def relative_l2_error_p1_3d(msh, w_hmm: np.ndarray, w_ref: np.ndarray) -> float:
    """The same norm quotient on a tetrahedral mesh: int w^2 = vol / 10 * (sum w_a^2 + sum_{a<b} w_a w_b)."""
    cv, vol = msh.cells, msh.cell_volumes()

    def sq(w):
        v = w[cv]                                                # [ne, 4]
        s1 = np.sum(v * v, axis=1)
        s2 = (np.sum(v, axis=1) ** 2 - s1) / 2.0
        return float(np.sum(vol / 10.0 * (s1 + s2)))

    return np.sqrt(sq(w_hmm - w_ref) / sq(w_ref))


def sample_p1(u: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Values of the fine P1 function u[(N+1), (N+1)] at the points pts[npoints, 2] (right-diagonal triangles)."""
    N = u.shape[0] - 1
    s = np.clip(pts * N, 0.0, N)
    i = np.minimum(s[:, 0].astype(int), N - 1)
    j = np.minimum(s[:, 1].astype(int), N - 1)
    a, b = s[:, 0] - i, s[:, 1] - j
    u00, u10, u01, u11 = u[i, j], u[i + 1, j], u[i, j + 1], u[i + 1, j + 1]
    lower = a >= b                    # triangle (v00, v10, v11) below the diagonal, (v00, v01, v11) above
    return np.where(lower, u00 + a * (u10 - u00) + b * (u11 - u10), u00 + b * (u01 - u00) + a * (u11 - u01))


def relative_l2_error_p1(msh, w_hmm: np.ndarray, w_ref: np.ndarray) -> float:
    """|| w_hmm - w_ref ||_L2 / || w_ref ||_L2 for two P1 functions given by nodal values on the triangular mesh msh (exact mass-matrix
    integration) -- calc_l2_error / calc_l2_norm of the reference's tests after interpolate_nonmatching."""
    cv = msh.cells                   # [n_cells, 3] vertex (= P1 dof) indices
    vol = msh.cell_volumes()

    def sq(w):
        a, b, c = w[cv[:, 0]], w[cv[:, 1]], w[cv[:, 2]]
        return float(np.sum(vol / 6.0 * (a * a + b * b + c * c + a * b + b * c + a * c)))

    return np.sqrt(sq(w_hmm - w_ref) / sq(w_ref))


def solve_fine_elasticity_2d(nx: int, ny: int, Lx: float, Ly: float, lam_of_x, mu_of_x, f_vec):
    """Nodal displacements u[(nx+1), (ny+1), 2] of the P1 vector solution of -div(C(x) : eps(u)) = f_vec (isotropic C from the Lame
    fields lam, mu) on [0, Lx] x [0, Ly], nx x ny rectangles split along the right diagonal, clamped at x = 0, traction-free elsewhere
    (test_integration_linear_elasticity.py:62-171).  Element means of lam, mu by the six-point rule; sparse direct solve."""
    hx, hy = Lx / nx, Ly / ny
    ix, iy = (a.ravel() for a in np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij"))
    nid = lambda i, j: i * (ny + 1) + j
    v00, v10, v01, v11 = nid(ix, iy), nid(ix + 1, iy), nid(ix, iy + 1), nid(ix + 1, iy + 1)
    tris = np.concatenate([np.stack([v00, v10, v11], 1), np.stack([v00, v01, v11], 1)])
    X = np.stack([np.repeat(np.arange(nx + 1) * hx, ny + 1), np.tile(np.arange(ny + 1) * hy, nx + 1)], 1)
    Pt = X[tris]                                                   # [ne, 3, 2]
    lam = np.zeros(len(tris))
    mu = np.zeros(len(tris))
    for q in range(6):
        xq = np.einsum("a,eac->ce", _BARY[q], Pt)
        lam += _WQ[q] * lam_of_x(xq)
        mu += _WQ[q] * mu_of_x(xq)
    d1, d2 = Pt[:, 1] - Pt[:, 0], Pt[:, 2] - Pt[:, 0]
    det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
    area = 0.5 * np.abs(det)
    g1 = np.stack([d2[:, 1], -d2[:, 0]], 1) / det[:, None]
    g2 = np.stack([-d1[:, 1], d1[:, 0]], 1) / det[:, None]
    G = np.stack([-g1 - g2, g1, g2], 1)                            # [ne, 3, 2] gradients of the three hat functions
    # strain of the basis function (vertex a, component c) in Voigt form (e_xx, e_yy, 2 e_xy)
    B = np.zeros((len(tris), 3, 6))
    for a in range(3):
        B[:, 0, 2 * a] = G[:, a, 0]
        B[:, 1, 2 * a + 1] = G[:, a, 1]
        B[:, 2, 2 * a] = G[:, a, 1]
        B[:, 2, 2 * a + 1] = G[:, a, 0]
    D = np.zeros((len(tris), 3, 3))
    D[:, 0, 0] = D[:, 1, 1] = lam + 2 * mu
    D[:, 0, 1] = D[:, 1, 0] = lam
    D[:, 2, 2] = mu
    Ke = area[:, None, None] * np.einsum("emi,emn,enj->eij", B, D, B)
    dofs = (2 * tris[:, :, None] + np.arange(2)[None, None, :]).reshape(len(tris), 6)
    nd = 2 * (nx + 1) * (ny + 1)
    K = sp.coo_matrix((Ke.ravel(), (np.repeat(dofs, 6, axis=1).ravel(), np.tile(dofs, (1, 6)).ravel())), shape=(nd, nd)).tocsr()
    b = np.zeros(nd)
    for c in range(2):
        b[c::2] = np.bincount(tris.ravel(), weights=np.repeat(area * f_vec[c] / 3.0, 3), minlength=nd // 2)
    clamped = np.zeros(nd, bool)
    cl_nodes = np.flatnonzero(np.isclose(X[:, 0], 0.0))
    clamped[2 * cl_nodes] = clamped[2 * cl_nodes + 1] = True
    free = np.flatnonzero(~clamped)
    S = sp.csr_matrix((np.ones(len(free)), (free, np.arange(len(free)))), shape=(nd, len(free)))
    u = np.zeros(nd)
    u[free] = spl.spsolve((S.T @ K @ S).tocsc(), S.T @ b, permc_spec="MMD_AT_PLUS_A", use_umfpack=False)
    return u.reshape(nx + 1, ny + 1, 2)


def sample_p1_rect(u: np.ndarray, pts: np.ndarray, Lx: float, Ly: float) -> np.ndarray:
    """sample_p1 for a nodal field u[(nx+1), (ny+1), ncomp] on [0, Lx] x [0, Ly]."""
    nx, ny = u.shape[0] - 1, u.shape[1] - 1
    sx = np.clip(pts[:, 0] / Lx * nx, 0.0, nx)
    sy = np.clip(pts[:, 1] / Ly * ny, 0.0, ny)
    i = np.minimum(sx.astype(int), nx - 1)
    j = np.minimum(sy.astype(int), ny - 1)
    a, b = (sx - i)[:, None], (sy - j)[:, None]
    u00, u10, u01, u11 = u[i, j], u[i + 1, j], u[i, j + 1], u[i + 1, j + 1]
    return np.where(a >= b, u00 + a * (u10 - u00) + b * (u11 - u10), u00 + b * (u01 - u00) + a * (u11 - u01))
