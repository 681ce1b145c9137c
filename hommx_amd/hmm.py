"""HMM solver classes with the constructor / ``solve()`` surface of flxrcz/hommx (``src/hommx/hmm.py``), driving the
batched HIP micro-cell solver instead of the per-cell DOLFINx/PETSc loop.

What is kept from the reference:
  * class names, constructor argument order, ``solve()``, ``set_boundary_conditions``, ``set_right_hand_side``,
    ``function_space`` (hmm.py:63-73, 173-176, 276-296, 434; class signatures :561-571, :717-728, :843-853, :976-987);
  * the log-don't-raise error convention (hmm.py:320-323, 427-430) and the ``_needs_reassembly`` cache (hmm.py:150, 287, 300-301, 332);
  * default homogeneous Dirichlet conditions for ``PoissonHMM`` (hmm.py:598-636), none for elasticity (hmm.py:806-807);
  * the macro algorithm of ``solve()`` (hmm.py:434-491): assemble, lift Dirichlet values bc by bc, zero rows+columns, solve.
What differs (DOLFINx/UFL/PETSc are not dependencies):
  * meshes are ``hommx_amd.mesh.Mesh``; ``A(x, y)``, ``f(x)``, ``Dtheta_transpose(x)`` are NumPy-vectorised callables
    (x: the cell midpoint, 3 components as the ``fem.Constant`` of hmm.py:190-192; y: array [dim, npts]);
  * the micro problems of ALL macro cells are solved in one batched GPU call (``_assemble_stiffness``);
  * the macro system is solved with a sparse direct solver (SciPy); ``petsc_options_*`` are accepted for signature
    compatibility and are advisory (a direct solve has no tolerance; the reference's own tightest test also uses LU,
    test_integration_poisson.py:207-212).
The micro solves have NO CPU path in this package: without the HIP library / a GPU, ``solve()`` raises.
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from collections.abc import Callable

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import fem
from .batch import MicroCellPlan
from .mesh import Mesh, micro_cells_per_side

_VOIGT = {2: [(0, 0), (1, 1), (0, 1)], 3: [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]}


def _unroll_dofs(dofs: np.ndarray, bs: int) -> np.ndarray:
    """hmm.py:31-40."""
    dofs = np.asarray(dofs)
    if bs == 1:
        return dofs
    return (dofs[..., None] * bs + np.arange(bs)).reshape(dofs.shape[:-1] + (-1,))


def micro_quadrature(dim: int, degree: int):
    """Barycentric points / weights of the rule that samples the coefficient: degree <= 1 centroid; 2: 3 / 4
    points; 3: 6-point Strang-Fix (triangle), 8-point collapsed Gauss-Jacobi (tetrahedron: positive weights -- the classical
    5-point Keast rule has a weight of -0.8 and can turn the element mean of a high-contrast coefficient non-positive; Basix's
    own 6-point Xiao-Gimbutas rule is not available offline: parity unpinned for that rule)."""
    import itertools

    if degree <= 1:
        return np.full((1, dim + 1), 1.0 / (dim + 1)), np.ones(1)
    if dim == 2 and degree == 2:
        p = np.full((3, 3), 1.0 / 6.0)
        np.fill_diagonal(p, 2.0 / 3.0)
        return p, np.full(3, 1.0 / 3.0)
    if dim == 2 and degree == 3:
        a, b, c = 0.659027622374092, 0.231933368553031, 0.109039009072877
        return np.array(list(itertools.permutations((a, b, c)))), np.full(6, 1.0 / 6.0)
    if dim == 3 and degree == 2:
        a, b = 0.5854101966249685, 0.1381966011250105
        p = np.full((4, 4), b)
        np.fill_diagonal(p, a)
        return p, np.full(4, 0.25)
    return fem.simplex_quadrature(dim, degree)


def hooke_to_voigt(C: np.ndarray, dim: int) -> np.ndarray:
    """[..., d,d,d,d] -> [..., t, t] with Cv[m][n] = E^m : C : E^n for the tensorial unit strains E^m."""
    pairs = _VOIGT[dim]
    t = len(pairs)
    out = np.empty(C.shape[:-4] + (t, t))
    for m, (i, j) in enumerate(pairs):
        for n, (k, l) in enumerate(pairs):
            out[..., m, n] = 0.25 * (C[..., i, j, k, l] + C[..., j, i, k, l] + C[..., i, j, l, k] + C[..., j, i, l, k])
    return out


class Lame:
    """Isotropic Hooke tensor given by its Lame parameters: ``A = lambda d_ij d_kl + mu (d_ik d_jl + d_il d_jk)``
    (test_integration_linear_elasticity.py:84-93; rotated_fibers.py:66-76).  Returning ``Lame(lam, mu)`` from the
    coefficient callable selects the 2-component isotropic kernel instead of the 21-component Voigt one."""

    def __init__(self, lam, mu):
        self.lam, self.mu = lam, mu


class TwoPhase:
    """Two-phase coefficient  A(x, y) = inside(x) if indicator(y) else outside(x)  -- the shape of every coefficient in the
    reference's examples (laminate.py:101-102, inclusion.py:107-118, rotated_fibers.py:23-38: ``ufl.conditional`` of the
    fast variable between two values).  Passing one as ``A`` lets the solver classes sample on the DEVICE: the phase mask is
    evaluated once on the micro mesh (element barycentres: UFL estimates degree 0 for a conditional, i.e. the centroid rule)
    and two values per macro cell are sent instead of n_el samples (``hommx_solve_batch_two_phase``).

    ``indicator(y)``: y[dim, npts] -> bool[npts];  ``inside(x)`` / ``outside(x)``: x[3, N_c] -> scalar, [N_c] array, or a
    ``Lame`` of such (elasticity).  The object is also a plain callable ``A(x, y)``, so every generic path accepts it.
    """

    def __init__(self, indicator, inside, outside):
        self.indicator, self.inside, self.outside = indicator, inside, outside

    @staticmethod
    def _pair(v, n):
        if isinstance(v, Lame):
            return np.stack([np.broadcast_to(np.asarray(v.lam, float), (n,)), np.broadcast_to(np.asarray(v.mu, float), (n,))], axis=-1)
        return np.broadcast_to(np.asarray(v, float), (n,))

    def phase_values(self, c: np.ndarray) -> np.ndarray:
        """[N_c, 2(, 2)]: (outside, inside) at the macro cell midpoints c[N_c, 3]."""
        n = c.shape[0]
        return np.stack([self._pair(self.outside(c.T), n), self._pair(self.inside(c.T), n)], axis=1)

    def __call__(self, x, y):
        m = np.asarray(self.indicator(y), dtype=bool)
        vin, vout = self.inside(x), self.outside(x)
        if isinstance(vin, Lame):
            return Lame(np.where(m, vin.lam, vout.lam), np.where(m, vin.mu, vout.mu))
        return np.where(m, vin, vout)


class Separable:
    """Separable scalar coefficient  A(x, y) = a(x) + b(x) g(y)  (``family="affine"``)  or  1 / (a(x) + b(x) g(y))
    (``family="reciprocal"``) -- the shape of the smooth coefficients in the reference's own tests
    (test_integration_poisson.py:124-125  1/(2 + cos 2 pi y0);  :149-150, 197  0.33 + 0.15 (sin 2 pi x0 + sin 2 pi y0);
    :268  1.1 + x0 + sin 2 pi y0).  Passing one as ``A`` of a Poisson solver class lets it sample on the DEVICE
    (``hommx_solve_batch_separable``): g is tabulated once on the micro mesh at the points of the degree-3 rule UFL would
    pick, and two numbers per macro cell cross the boundary instead of n_el samples.

    ``a(x)`` / ``b(x)``: x[3, N_c] -> scalar or [N_c];  ``g(y)``: y[dim, npts] -> [npts].  For the elasticity classes ``a`` / ``b`` return
    ``Lame(lam, mu)`` (affine family): lambda = a.lam + b.lam g, mu = a.mu + b.mu g -- the isotropic Hooke tensor of the reference's 2D
    beam test, lambda = 1.25, mu = 5 + 4.5 sin 2 pi y0 (test_integration_linear_elasticity.py:78-93).  The object is also a plain
    callable ``A(x, y)``.  ``host_stream`` is the documented host equivalent of the device sampler: the same IEEE operations in the
    same order, hence the same bits (tests/test_gpu_separable.py)."""

    def __init__(self, family: str, a, b, g, degree: int = 3):
        """``degree``: quadrature degree of the rule that samples g -- 3 is UFL's estimate for ONE transcendental function of
        the fast variable (sin 2 pi y0: 1 + 2), the shape of every smooth coefficient in the reference's tests."""
        if family not in ("affine", "reciprocal"):
            raise ValueError("family must be 'affine' or 'reciprocal'")
        self.family, self.a, self.b, self.g, self.degree = family, a, b, g, int(degree)

    def params(self, c: np.ndarray) -> np.ndarray:
        """[N_c, 2] = (a, b) at the macro cell midpoints c[N_c, 3]; [N_c, 2, 2] = ((a, b) of lambda, (a, b) of mu) when a / b return ``Lame``."""
        n = c.shape[0]
        a, b = self.a(c.T), self.b(c.T)
        bc = lambda v: np.broadcast_to(np.asarray(v, float), (n,))
        if isinstance(a, Lame) or isinstance(b, Lame):
            a = a if isinstance(a, Lame) else Lame(a, a)
            b = b if isinstance(b, Lame) else Lame(b, b)
            return np.stack([np.stack([bc(a.lam), bc(b.lam)], axis=1), np.stack([bc(a.mu), bc(b.mu)], axis=1)], axis=1)
        return np.stack([bc(a), bc(b)], axis=1)

    def table(self, yq: np.ndarray, w: np.ndarray) -> np.ndarray:
        """yq[n_el, n_q, dim] -> what the C ABI takes: affine: element means of g [n_el]; reciprocal: g at the points [n_el, n_q]."""
        n_el, nq, d = yq.shape
        gq = np.asarray(self.g(yq.reshape(-1, d).T), float).reshape(n_el, nq)
        if self.family == "affine":
            acc = np.zeros(n_el)
            for q in range(nq):
                acc = acc + w[q] * gq[:, q]
            return acc
        return gq

    def host_stream(self, params: np.ndarray, table: np.ndarray, w: np.ndarray) -> np.ndarray:
        """Element means coef[N_c, n_el(, 2)] exactly as the device sampler forms them."""
        if params.ndim == 3:  # one (a, b) pair per Lame parameter
            return np.stack([self.host_stream(params[:, k], table, w) for k in range(params.shape[1])], axis=-1)
        a, b = params[:, 0:1], params[:, 1:2]
        if self.family == "affine":
            return a + b * table[None, :]
        acc = np.zeros((params.shape[0], table.shape[0]))
        for q in range(table.shape[1]):
            acc = acc + w[q] * (1.0 / (a + b * table[None, :, q]))
        return acc

    def __call__(self, x, y):
        a, b, g = self.a(x), self.b(x), self.g(y)
        if isinstance(a, Lame) or isinstance(b, Lame):
            if self.family != "affine":
                raise ValueError("Lame-valued Separable coefficients are affine (lambda, mu = a + b g)")
            a = a if isinstance(a, Lame) else Lame(a, a)
            b = b if isinstance(b, Lame) else Lame(b, b)
            return Lame(a.lam + b.lam * g, a.mu + b.mu * g)
        v = a + b * g
        return v if self.family == "affine" else 1.0 / v


def isotropic_hooke(lam, mu, dim: int) -> np.ndarray:
    lam, mu = np.asarray(lam, float), np.asarray(mu, float)
    I = np.eye(dim)
    t1 = np.einsum("ij,kl->ijkl", I, I)
    t2 = np.einsum("ik,jl->ijkl", I, I) + np.einsum("il,jk->ijkl", I, I)
    return lam[..., None, None, None, None] * t1 + mu[..., None, None, None, None] * t2


class BaseHMM(ABC):
    """hmm.py:53-511."""

    _kind = "poisson"

    def __init__(
        self,
        msh: Mesh,
        A: Callable,
        f: Callable,
        msh_micro: Mesh,
        eps: float,
        petsc_options_global_solve: dict | None = None,
        petsc_options_cell_problem: dict | None = None,
        petsc_options_prefix: str = "hommx_HMM",
        *,
        quadrature_degree: int | None = None,
        rhs_quadrature_degree: int = 6,
        device: int | None = None,
        reserve: bool = False,
    ):
        """``quadrature_degree``: degree of the micro quadrature rule that samples ``A(x, .)``.  The reference lets UFL estimate it
        from the expression (hmm.py:190-198 + fem.form at :644-647): 0 for a ``conditional`` between constants, 3 for one
        transcendental function of the fast variable -- ``sin(2 pi y0)``, ``1/(2 + cos(2 pi y0))``: the coefficients of its own
        tests (test_integration_poisson.py:124-125, 149-150, 197).  A Python callable has no expression tree, so the default
        ``None`` is a HEURISTIC, not UFL's estimate (a product of two transcendental factors would be 6 there, a polynomial its own
        degree): it looks at the samples of the first, the middle and the last macro cell -- piecewise constant on every micro
        element, or only a handful of distinct values over the cell -> centroid rule (degree 0), anything else -> degree 3 --
        logs the choice with its evidence at WARNING level and raises when the three cells disagree.  Pass an integer to choose
        the rule yourself; ``TwoPhase`` / ``Separable`` coefficients carry their own degree and never guess.
        ``device``: HIP device ordinal; resolved at the first solve (``dist.default_device``: the ordinal given to
        ``dist.select_device``, the process group's bound device, the torch device the caller selected, else LOCAL_RANK modulo the
        visible devices, else 0).
        ``reserve``: create the plan and allocate its device workspace for this rank's share of the macro cells NOW (``prepare()``)
        instead of inside the first ``solve()`` -- the nested-dissection route of the C4 / C5 size holds 0.2 GB of fronts per cell
        of a chunk, 0.5 - 3 s of ``hipMalloc`` that the first assembly would otherwise hide.  Resolves the device at construction
        time, so leave it off when the solver is built before ``init_process_group``; ``prepare()`` can be called later."""
        self._logger = logging.getLogger(__name__)
        self._msh = msh
        self._comm = msh.comm
        self._coeff = A
        self._f = f
        self._eps = eps
        self._cell_mesh = msh_micro
        self._tdim = msh.topology.dim
        if self._tdim not in (2, 3):
            raise ValueError("Topology should be 3D or 2D")  # hmm.py:104-105
        if self._tdim != msh.geometry.dim:
            raise ValueError(
                "Topological dimension is different from geometrical dimension. Currently surfaces in 3D are not supported."
            )
        if msh_micro.topology.dim != msh_micro.geometry.dim:
            raise ValueError("Topological dimension is different from geometrical dimension for micro mesh.")
        if self._tdim != msh_micro.topology.dim:
            raise ValueError("Micro and macro mesh should have the same dimensionality.")  # hmm.py:114-115
        self._n_micro = micro_cells_per_side(msh_micro)
        self._cell_mesh_area = float(msh_micro.cell_volumes().sum())  # hmm.py:101
        if isinstance(A, TwoPhase) and quadrature_degree not in (None, 0, 1):
            raise ValueError("a TwoPhase coefficient is piecewise constant in y (UFL: degree 0, centroid rule); "
                             f"quadrature_degree={quadrature_degree} would be ignored by the device sampler")
        self._quadrature_degree = quadrature_degree  # None: decided from the samples on first use (quadrature_degree_used)
        self.quadrature_degree_used: int | None = 0 if isinstance(A, TwoPhase) else quadrature_degree
        if isinstance(A, Separable) and quadrature_degree is None:
            self.quadrature_degree_used = A.degree
        self._rhs_degree = rhs_quadrature_degree
        self._device = device  # None: resolved lazily in _ensure_plan (the solver may be built before init_process_group)
        self._reserve_at_construction = bool(reserve)

        self._V_macro = self._setup_macro_function_space()
        self._macro_coordinates = self._V_macro.tabulate_dof_coordinates()
        self._bs = self._V_macro.bs
        self._num_basis_functions_per_cell = (self._tdim + 1) * self._bs  # hmm.py:138-140
        self._num_global_dofs = self._V_macro.num_dofs
        self._u = fem.Function(self._V_macro)
        self._A = None
        self._needs_reassembly = True
        if petsc_options_cell_problem is None:
            petsc_options_cell_problem = {"ksp_atol": 1e-10}  # hmm.py:153-155 (advisory here)
        self._petsc_options_cell_problem = petsc_options_cell_problem
        self._petsc_options_global_solve = petsc_options_global_solve
        self._petsc_options_prefix = petsc_options_prefix
        self._bcs: list[fem.DirichletBC] = []
        self._Dtheta_t = None
        self._plan: MicroCellPlan | None = None
        self.effective_tensors: np.ndarray | None = None  # A_H / C_H of every macro cell after assembly
        self.cell_info: np.ndarray | None = None
        if self._reserve_at_construction:
            self.prepare()

    # -- API ---------------------------------------------------------------------------------------
    @property
    def function_space(self) -> fem.FunctionSpace:
        """Function space of the macro mesh (hmm.py:173-176)."""
        return self._V_macro

    def set_boundary_conditions(self, bcs):
        """hmm.py:276-287."""
        self._bcs = bcs if isinstance(bcs, list) else [bcs]
        self._needs_reassembly = True

    def set_right_hand_side(self, f: Callable):
        """hmm.py:289-296."""
        self._f = f

    @abstractmethod
    def _setup_macro_function_space(self) -> fem.FunctionSpace: ...

    # -- coefficient sampling (hmm.py:190-198, 349-352) ---------------------------------------------
    def _sample_one(self, c_T: np.ndarray, yq: np.ndarray):
        """A(c_T, y) at the quadrature points yq[dim, npts] -> array [npts, ...] (or Lame)."""
        v = self._coeff(c_T, yq)
        if isinstance(v, Lame):
            lam = np.broadcast_to(np.asarray(v.lam, float), (yq.shape[1],))
            mu = np.broadcast_to(np.asarray(v.mu, float), (yq.shape[1],))
            return np.stack([lam, mu], axis=-1)
        v = np.asarray(v, dtype=float)
        if v.ndim == 0 or v.shape[0] != yq.shape[1]:
            v = np.broadcast_to(v, (yq.shape[1],) + v.shape).copy()
        return v

    def _quadrature_evidence(self, c_T: np.ndarray) -> tuple[int, str]:
        """Degree the heuristic picks from the samples of ONE macro cell, and why."""
        d = self._tdim
        bary, _ = micro_quadrature(d, 3)
        Xe = self._cell_mesh.cell_vertices()
        yq = np.einsum("qa,eak->eqk", bary, Xe)
        v = self._sample_one(c_T, yq.reshape(-1, d).T)
        v = v.reshape((yq.shape[0], yq.shape[1]) + v.shape[1:])
        if np.all(v == v[:, :1]):
            return 0, "constant on every micro element"
        flat = v.reshape(v.shape[0] * v.shape[1], -1)
        distinct = max(int(np.unique(flat[:, k]).size) for k in range(flat.shape[1]))
        if distinct <= 8:
            return 0, f"{distinct} distinct values over the cell (a conditional whose interface cuts elements)"
        return 3, f"{distinct} distinct values over the cell, varying inside elements (smooth in y)"

    def _auto_quadrature_degree(self, c_T: np.ndarray | None = None) -> int:
        """Default policy (see __init__): sample A(c_T, .) at the degree-3 points of every micro element of the first, the middle
        and the last macro cell.  A ``conditional`` between constants (UFL: degree 0, centroid rule) shows up as a coefficient
        that is constant on every element or -- when its interface cuts through elements, like the wrapped disc of
        inclusion.py:107-118 -- takes only a handful of distinct values over the whole cell; anything else is treated as smooth:
        degree 3.  The choice is logged with its evidence; cells that disagree raise (a wrong guess would silently solve a
        different discrete problem from the reference's)."""
        mid = self._msh.cell_midpoints()
        probe = [0, len(mid) // 2, len(mid) - 1] if c_T is None else [None]
        found = [self._quadrature_evidence(mid[k] if k is not None else c_T) for k in probe]
        degrees = sorted({f[0] for f in found})
        if len(degrees) > 1:
            raise ValueError("cannot guess the micro quadrature degree of A(x, y): macro cells " + ", ".join(
                f"{k}: degree {f[0]} ({f[1]})" for k, f in zip(probe, found)) + " -- pass quadrature_degree= explicitly "
                "(0 for a conditional between constants, 3 for one transcendental function of y: what UFL estimates, hmm.py:190-198)")
        self._logger.warning("quadrature_degree not given: guessed degree %d for A(x, y) from its samples (%s); pass "
                             "quadrature_degree= to choose the rule UFL would estimate for your expression",
                             degrees[0], "; ".join(f"cell {k}: {f[1]}" for k, f in zip(probe, found)))
        return degrees[0]

    def _element_means(self, cells: np.ndarray) -> tuple[np.ndarray, str]:
        d = self._tdim
        if self.quadrature_degree_used is None:
            self.quadrature_degree_used = self._auto_quadrature_degree()
        bary, w = micro_quadrature(d, self.quadrature_degree_used)
        Xe = self._cell_mesh.cell_vertices()  # [n_el, d+1, d]
        yq = np.einsum("qa,eak->eqk", bary, Xe)
        n_el, nq = yq.shape[:2]
        yflat = yq.reshape(-1, d).T
        c = self._msh.cell_midpoints()[cells]
        out = self._sample_batched(c, yflat, n_el, nq, w)
        if out is None:  # the callable is not broadcastable over cells: one call per macro cell (as the reference)
            first = self._sample_one(c[0], yflat)
            out = np.empty((len(cells),) + (n_el,) + first.shape[1:])
            for k in range(len(cells)):
                v = first if k == 0 else self._sample_one(c[k], yflat)
                out[k] = np.tensordot(w, v.reshape((n_el, nq) + v.shape[1:]), axes=([0], [1]))
        return self._pack_coefficient(out)

    def _sample_batched(self, c: np.ndarray, yflat: np.ndarray, n_el: int, nq: int, w: np.ndarray):
        """Try ONE broadcast call A(x[3, N_c, 1], y[d, 1, npts]) -> [N_c, npts(, ...)] per chunk of cells; accept it only
        if it reproduces the per-cell call on the first and last cell.  Returns None when the callable cannot broadcast."""
        nc, npts = c.shape[0], yflat.shape[1]
        if nc < 4:
            return None
        try:
            ref0 = self._sample_one(c[0], yflat)
            ref1 = self._sample_one(c[-1], yflat)
            chunk = max(1, int(2.0e7 // max(1, npts)))
            parts = []
            for a in range(0, nc, chunk):
                xb = c[a : a + chunk].T[:, :, None]
                v = self._coeff(xb, yflat[:, None, :])
                if isinstance(v, Lame):
                    lam = np.broadcast_to(np.asarray(v.lam, float), (xb.shape[1], npts))
                    mu = np.broadcast_to(np.asarray(v.mu, float), (xb.shape[1], npts))
                    v = np.stack([lam, mu], axis=-1)
                else:
                    v = np.asarray(v, dtype=float)
                    if v.ndim < 2 or v.shape[:2] != (xb.shape[1], npts):
                        if v.ndim >= 2 and v.shape[:2] == (1, npts) or v.ndim == 0:
                            v = np.broadcast_to(v, (xb.shape[1], npts) + v.shape[2:]) if v.ndim else np.full((xb.shape[1], npts), float(v))
                        else:
                            return None
                if nq == 1:  # centroid rule: the mean IS the sample (no copy)
                    parts.append(v.reshape((v.shape[0], n_el) + v.shape[2:]))
                else:
                    parts.append(np.tensordot(v.reshape((v.shape[0], n_el, nq) + v.shape[2:]), w, axes=([2], [0])))
            out = parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)
            chk0 = np.tensordot(w, ref0.reshape((n_el, nq) + ref0.shape[1:]), axes=([0], [1]))
            chk1 = np.tensordot(w, ref1.reshape((n_el, nq) + ref1.shape[1:]), axes=([0], [1]))
            if out[0].shape != chk0.shape or not (np.allclose(out[0], chk0, rtol=1e-14, atol=0) and np.allclose(out[-1], chk1, rtol=1e-14, atol=0)):
                return None
            return out
        except Exception:
            return None

    def _pack_coefficient(self, means: np.ndarray) -> tuple[np.ndarray, str]:
        """Element means -> the layout of include/hommx_hip.h for the plan kind."""
        d = self._tdim
        if self._kind == "poisson":
            if means.ndim == 2:
                return means, "poisson"
            if means.shape[-2:] == (d, d):
                asym = np.abs(means - np.swapaxes(means, -1, -2)).max() if means.size else 0.0
                if asym > 1e-12 * max(1.0, float(np.abs(means).max())):
                    raise ValueError("matrix-valued A must be symmetric: the kernels form the Schur complement C0 - B^T K^+ B, which "
                                     "equals the reference's energy functional (hmm.py:652-667) only for symmetric A "
                                     f"(max |A - A^T| = {asym:.3e})")
                pairs = _VOIGT[d]
                return np.stack([0.5 * (means[..., i, j] + means[..., j, i]) for i, j in pairs], axis=-1), "poisson_matrix"
            raise ValueError(f"PoissonHMM coefficient must be scalar or {d}x{d}; got trailing shape {means.shape[2:]}")
        if means.ndim == 3 and means.shape[-1] == 2:
            return means, "elasticity"
        if means.shape[-4:] == (d, d, d, d):
            cv = hooke_to_voigt(means, d)
            t = cv.shape[-1]
            iu = np.triu_indices(t)
            return cv[..., iu[0], iu[1]], "elasticity_voigt"
        raise ValueError("elasticity coefficient must be Lame(lam, mu) or a [d,d,d,d] tensor")

    def _stratification(self, cells: np.ndarray) -> np.ndarray | None:
        """M[c] = Dtheta_transpose(c_T) of the given macro cells (hmm.py:756-757, 1015-1016).  ONE broadcast call
        ``Dtheta_transpose(x[3, N_c]) -> [d, d, N_c]`` when the callable allows it (accepted only if it reproduces the per-cell
        call on the first and the last cell), else one call per cell as the reference does."""
        if self._Dtheta_t is None:
            return None
        c = self._msh.cell_midpoints()[cells]
        d = self._tdim
        nc = len(cells)

        def one(x):
            m = np.asarray(self._Dtheta_t(x), dtype=float)
            if m.shape != (d, d):
                raise ValueError(f"Dtheta_transpose must return a {d}x{d} matrix (hmm.py:741, :762); got {m.shape}")
            return m

        if nc == 0:
            return np.empty((0, d, d))
        first, last = one(c[0]), one(c[-1])
        known = {0: first, nc - 1: last}
        if nc >= 4:
            # the broadcast result is accepted only if it reproduces the per-cell call (what the reference does, hmm.py:756-757) on the
            # first, the middle, the last and a few more cells spread over the batch -- a callable that broadcasts but reduces over its
            # argument, or indexes into it, agrees at the ends at best
            probe = sorted({nc // 2, nc // 3, (2 * nc) // 3, (7 * nc) // 11, 1, nc - 2} - {0, nc - 1})
            try:
                rows = self._Dtheta_t(c.T)
                mb = np.empty((d, d, nc))
                for i in range(d):
                    for j in range(d):
                        mb[i, j] = np.broadcast_to(np.asarray(rows[i][j], dtype=float), (nc,))
                mb = np.ascontiguousarray(np.moveaxis(mb, -1, 0))
                for k in probe:
                    known[k] = one(c[k])
                if all(np.array_equal(mb[k], v) for k, v in known.items()):
                    return mb
                self._logger.debug("Dtheta_transpose: the broadcast call disagrees with the per-cell calls; evaluating cell by cell")
            except Exception as exc:
                self._logger.debug(f"Dtheta_transpose: no broadcast evaluation ({type(exc).__name__}: {exc}); evaluating cell by cell")
        M = np.empty((nc, d, d))
        for k in range(nc):
            M[k] = known[k] if k in known else one(c[k])
        return M

    # -- the hot path (replaces the loop hmm.py:298-332) ---------------------------------------------
    @staticmethod
    def _sharded() -> bool:
        """Only shard when the caller already runs under an initialised torch.distributed group with more than one rank."""
        import sys

        dist = sys.modules.get("torch.distributed")
        return dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _ensure_plan(self, kind: str, n_cells: int | None = None) -> MicroCellPlan:
        """The plan of this solver (created on first use: replaces the per-solve object construction of hmm.py:420-425).  With
        ``n_cells`` (``prepare()`` / ``reserve=True``) the plan's device workspace is allocated for batches of that size right away
        (``hommx_plan_reserve``); a solve without it allocates what its first batch needs (half as much on the C4 / C5 route: a caller
        who pays ahead of many batches gets the bigger chunks, one who pays inside the only solve does not want to)."""
        if self._plan is None or self._plan.kind != kind:
            if self._device is None:
                from .dist import default_device

                self._device = default_device()
            self._plan = MicroCellPlan(self._tdim, self._n_micro, kind, device=self._device)
            self._reserved_cells = 0
        if n_cells and n_cells > getattr(self, "_reserved_cells", 0) and hasattr(self._plan, "reserve"):
            self._plan.reserve(int(n_cells))
            self._reserved_cells = int(n_cells)
        return self._plan

    def _local_cell_count(self) -> int:
        """Macro cells this rank solves: all of them, or its block under a process group (hmm.py:307-310)."""
        n = self._msh.num_cells
        if self._sharded():
            import torch.distributed as dist

            from .dist import shard_range

            b, e, _ = shard_range(n, dist.get_rank(), dist.get_world_size())
            return e - b
        return n

    def prepare(self) -> "BaseHMM":
        """Create the plan and allocate its device workspace for this rank's share of the macro cells now, so that the first
        ``solve()`` does not pay for it (not in the reference's API: there every PETSc object is made per solve, hmm.py:420-425).
        The kernel family depends on the coefficient's shape, which a plain callable only shows when sampled: one macro cell is
        sampled for that."""
        if isinstance(self._coeff, (TwoPhase, Separable)):
            kind = "poisson" if self._kind == "poisson" else "elasticity"
            if isinstance(self._coeff, Separable) and self._kind != "poisson" and self._coeff.family != "affine":
                kind = self._element_means(np.array([0]))[1]
        else:
            kind = self._element_means(np.array([0]))[1]
        self._ensure_plan(kind, self._local_cell_count())
        return self

    def _shard_device(self):
        """Device of the gather buffer under RCCL = the plan's device (None before a plan exists on a gloo / stub-plan run)."""
        return getattr(self._plan, "device", None) if self._plan is not None else self._device

    def _tensor_size(self) -> int:
        d = self._tdim
        return d if self._kind == "poisson" else d * (d + 1) // 2

    def _effective_tensors(self, cells: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """A_H / C_H and the per-cell info flags of ``cells``.  Under a process group every rank samples, uploads and solves
        ONLY its block of the cells (the reference's MPI partition, hmm.py:307-310); one all-gather returns the whole field and
        the real info vector to every rank (hommx_amd/dist.py)."""
        if isinstance(self._coeff, TwoPhase):
            res = self._effective_tensors_two_phase(cells)
            if res is not None:
                return res
        if isinstance(self._coeff, Separable) and (self._kind == "poisson" or self._coeff.family == "affine"):
            res = self._effective_tensors_separable(cells)
            if res is not None:
                return res
        if self._sharded():
            from .dist import run_sharded, solve_block

            def local(b, e):
                sub = cells[b:e]
                coef, kind = self._element_means(sub)
                return solve_block(self._ensure_plan(kind), coef, self._stratification(sub))

            return run_sharded(self._tensor_size(), len(cells), local, device=self._shard_device())
        coef, kind = self._element_means(cells)
        M = self._stratification(cells)
        return self._ensure_plan(kind).solve(coef, M, return_info=True)

    def _effective_tensors_separable(self, cells: np.ndarray):
        """Device-side sampling of a ``Separable`` coefficient: one table of g on the micro mesh + (a, b) per macro cell (per Lame
        parameter for the elasticity classes: test_integration_linear_elasticity.py:78-93)."""
        plan = self._ensure_plan("poisson" if self._kind == "poisson" else "elasticity")
        if not hasattr(plan, "solve_separable"):
            return None
        d = self._tdim
        if self.quadrature_degree_used is None:  # not reached for Separable (it carries its degree); kept for subclasses
            self.quadrature_degree_used = self._auto_quadrature_degree()
        bary, w = micro_quadrature(d, self.quadrature_degree_used)
        yq = np.einsum("qa,eak->eqk", bary, self._cell_mesh.cell_vertices())
        co = self._coeff
        table = co.table(yq, w)

        if self._sharded():
            from .dist import run_sharded, solve_block_separable

            return run_sharded(self._tensor_size(), len(cells),
                               lambda b, e: solve_block_separable(plan, co.family, table, w, co.params(self._msh.cell_midpoints()[cells[b:e]]),
                                                                  self._stratification(cells[b:e])), device=self._shard_device())
        return plan.solve_separable(co.family, table, w, co.params(self._msh.cell_midpoints()[cells]), self._stratification(cells),
                                    return_info=True)

    def _effective_tensors_two_phase(self, cells: np.ndarray):
        """Device-side sampling of a ``TwoPhase`` coefficient: one mask + two values per macro cell."""
        kind = "poisson" if self._kind == "poisson" else "elasticity"
        plan = self._ensure_plan(kind)
        if not hasattr(plan, "solve_two_phase"):
            return None
        d = self._tdim
        yb = self._cell_mesh.cell_midpoints()[:, :d].T  # element barycentres
        mask = np.asarray(self._coeff.indicator(yb), dtype=bool)

        def values_of(sub):
            v = self._coeff.phase_values(self._msh.cell_midpoints()[sub])
            if (v.ndim == 2) != (kind == "poisson"):
                raise ValueError("TwoPhase values must be scalars for PoissonHMM and Lame(lam, mu) for LinearElasticityHMM")
            return v

        if self._sharded():
            from .dist import run_sharded, solve_block_two_phase

            return run_sharded(self._tensor_size(), len(cells),
                               lambda b, e: solve_block_two_phase(plan, mask, values_of(cells[b:e]), self._stratification(cells[b:e])),
                               device=self._shard_device())
        return plan.solve_two_phase(mask, values_of(cells), self._stratification(cells), return_info=True)

    def _local_stiffness_from_tensors(self, cells: np.ndarray, AH: np.ndarray) -> np.ndarray:
        """S_loc = vol(T)/vol(Y) * (macro gradients) A_H (macro gradients)^T  == hmm.py:361-369."""
        d, bs = self._tdim, self._bs
        X = self._msh.cell_vertices()[cells]  # [nc, d+1, d]
        ones = np.ones(X.shape[:2] + (1,))
        Minv = np.linalg.inv(np.concatenate([ones, X], axis=2))
        G = np.transpose(Minv[:, 1:, :], (0, 2, 1))  # [nc, a, d]  grad phi_a
        vol = self._msh.cell_volumes()[cells] / self._cell_mesh_area
        if bs == 1:
            return vol[:, None, None] * np.einsum("cai,cij,cbj->cab", G, AH, G)
        pairs = _VOIGT[d]
        I = np.eye(d)
        eps = 0.5 * (np.einsum("pi,caj->capij", I, G) + np.einsum("pj,cai->capij", I, G))
        eps = eps.reshape(len(cells), (d + 1) * bs, d, d)
        Wv = np.stack([eps[:, :, k, l] * (1.0 if k == l else 2.0) for (k, l) in pairs], axis=-1)
        return vol[:, None, None] * np.einsum("cam,cmn,cbn->cab", Wv, AH, Wv)

    def _compute_local_stiffness(self, cell_index: int) -> np.ndarray:
        """Single-cell form of the reference seam (hmm.py:334-369): a batch of one."""
        cells = np.array([cell_index])
        AH, info = self._effective_tensors(cells)
        return self._local_stiffness_from_tensors(cells, AH)[0]

    def _periodic_to_micro_nodes(self) -> np.ndarray:
        """Micro-mesh node -> periodic unknown (the slave -> master map of cell_problem.py:38-300 on the torus)."""
        n, d = self._n_micro, self._tdim
        g = np.rint(self._cell_mesh.geometry.x[:, :d] * n).astype(np.int64) % n
        return g @ (n ** np.arange(d))

    def correctors_for_cell(self, cell_index: int) -> list[fem.Function]:
        """The nb correctors of one macro cell as functions on the micro mesh -- what the reference leaves in
        ``self._correctors`` after ``_compute_local_stiffness(cell_index)`` (hmm.py:204-207, 354-358, 431).

        Computed from the canonical correctors chi_m returned by the GPU: the macro basis functions are affine on the
        sampling box, so corrector_i = eps * sum_m w_m(grad phi_i) chi_m, up to the additive constant
        the reference's Krylov solve leaves undetermined (returned mean-free)."""
        cells = np.array([cell_index])
        coef, kind = self._element_means(cells)
        M = self._stratification(cells)
        _, chi = self._ensure_plan(kind).solve(coef, M, return_correctors=True)  # [1, t, n^d * bs]
        d, bs = self._tdim, self._bs
        X = self._msh.cell_vertices()[cells][0]
        G = np.linalg.inv(np.concatenate([np.ones((d + 1, 1)), X], axis=1))[1:, :].T  # grad phi_a
        if bs == 1:
            Wv = G  # [nb, t = d]
        else:
            I = np.eye(d)
            eps_ = 0.5 * (np.einsum("pi,aj->apij", I, G) + np.einsum("pj,ai->apij", I, G)).reshape((d + 1) * bs, d, d)
            Wv = np.stack([eps_[:, k, l] * (1.0 if k == l else 2.0) for (k, l) in _VOIGT[d]], axis=-1)
        per = self._eps * (Wv @ chi[0])  # [nb, n^d * bs]
        idx = self._periodic_to_micro_nodes()
        V_micro = fem.FunctionSpace(self._cell_mesh, bs)
        out = []
        for i in range(per.shape[0]):
            f = fem.Function(V_micro)
            f.x.array[:] = per[i].reshape(-1, bs)[idx].ravel()
            out.append(f)
        return out

    def _assemble_stiffness(self):
        """hmm.py:298-332 with the cell loop replaced by one batched call."""
        if not self._needs_reassembly:
            return
        cells = np.arange(self._msh.num_cells)
        AH, info = self._effective_tensors(cells)
        S = self._local_stiffness_from_tensors(cells, AH)
        bad = np.nonzero((info != 0) | np.isnan(S).any(axis=(1, 2)))[0]
        for c in bad:  # hmm.py:320-323: log, do not raise
            self._logger.error(f"Something went wrong when calculating local matrix on cell {c}")
        dofs = _unroll_dofs(self._msh.cells.astype(np.int64), self._bs)  # [nc, nb]
        nb = dofs.shape[1]
        rows = np.repeat(dofs, nb, axis=1).ravel()
        cols = np.tile(dofs, (1, nb)).ravel()
        N = self._num_global_dofs
        self._A = sp.coo_matrix((S.ravel(), (rows, cols)), shape=(N, N)).tocsr()  # MatSetValues(ADD), hmm.py:325-330
        self.effective_tensors, self.cell_info = AH, info
        self._needs_reassembly = False

    def solve(self) -> fem.Function:
        """Assemble the LHS, RHS and solve the problem (hmm.py:434-491)."""
        self._assemble_stiffness()
        A = self._A.copy()
        b = fem.assemble_load_vector(self._V_macro, self._f, self._rhs_degree)
        for bc in self._bcs:  # hmm.py:453-480, bc by bc
            idx, val = bc.unrolled()
            u_bc = np.zeros(self._num_global_dofs)
            u_bc[idx] = val
            b_lift = A @ u_bc
            b[idx] = val
            b -= b_lift
            A = _zero_rows_columns(A, idx)
            b[idx] = val
        # the reference caches the BC-modified matrix; keep the clean one and re-apply -- same result
        if np.isnan(b).any() or np.isnan(A.data).any():
            self._logger.error("Something went wrong in the global problem solve. NaN in the assembled system")
        x = spla.spsolve(A.tocsc(), b)
        self._u.x.array[:] = x
        return self._u

    def plot_solution(self, u=None):  # hmm.py:493-511 (visualisation: out of scope)
        raise NotImplementedError("plotting is out of scope of hommx_amd; use u.x.array with any plotting tool")


def _zero_rows_columns(A: sp.csr_matrix, idx: np.ndarray) -> sp.csr_matrix:
    """PETSc MatZeroRowsColumns(idx, diag=1.0) (hmm.py:478)."""
    n = A.shape[0]
    keep = np.ones(n)
    keep[idx] = 0.0
    Dk = sp.diags(keep)
    A = Dk @ A @ Dk
    return (A + sp.diags(1.0 - keep)).tocsr()


def _box_boundary_nodes(msh: Mesh) -> np.ndarray:
    x = msh.geometry.x[:, : msh.topology.dim]
    lo, hi = x.min(axis=0), x.max(axis=0)
    return np.nonzero(np.any(np.isclose(x, lo) | np.isclose(x, hi), axis=1))[0]


class PoissonHMM(BaseHMM):
    """hmm.py:514-667.  Forms: micro LHS :644-647, RHS :649-650, local stiffness :652-667."""

    _kind = "poisson"

    def __init__(self, msh, A, f, msh_micro, eps, petsc_options_global_solve=None, petsc_options_cell_problem=None,
                 petsc_options_prefix: str = "hommx_PoissonHMM", **kw):
        super().__init__(msh, A, f, msh_micro, eps, petsc_options_global_solve, petsc_options_cell_problem,
                         petsc_options_prefix, **kw)
        # homogeneous Dirichlet on the bounding box by default (hmm.py:598-636)
        self._bcs = [fem.dirichletbc(0.0, _box_boundary_nodes(self._msh), self._V_macro)]

    def _setup_macro_function_space(self):
        return fem.functionspace(self._msh, ("Lagrange", 1))


class PoissonStratifiedHMM(PoissonHMM):
    """hmm.py:670-789: ``Dtheta_transpose`` comes right after ``eps`` (hmm.py:717-728); M[i][j] = d theta_j/d x_i."""

    def __init__(self, msh, A, f, msh_micro, eps, Dtheta_transpose, petsc_options_global_solve=None,
                 petsc_options_cell_problem=None, petsc_options_prefix: str = "hommx_PoissonStratifiedHMM", **kw):
        super().__init__(msh, A, f, msh_micro, eps, petsc_options_global_solve, petsc_options_cell_problem,
                         petsc_options_prefix, **kw)
        self._Dtheta_t = Dtheta_transpose


class LinearElasticityHMM(BaseHMM):
    """hmm.py:792-922.  No boundary condition by default (hmm.py:806-807)."""

    _kind = "elasticity"

    def __init__(self, msh, A, f, msh_micro, eps, petsc_options_global_solve=None, petsc_options_cell_problem=None,
                 petsc_options_prefix: str = "hommx_LinearElasticityHMM", **kw):
        super().__init__(msh, A, f, msh_micro, eps, petsc_options_global_solve, petsc_options_cell_problem,
                         petsc_options_prefix, **kw)

    def _setup_macro_function_space(self):
        return fem.functionspace(self._msh, ("Lagrange", 1, (self._tdim,)))


class LinearElasticityStratifiedHMM(LinearElasticityHMM):
    """hmm.py:925-1067: e_D(u) = sym(Dtheta^T . nabla_grad u) (:1024-1030)."""

    def __init__(self, msh, A, f, msh_micro, eps, Dtheta_transpose, petsc_options_global_solve=None,
                 petsc_options_cell_problem=None, petsc_options_prefix: str = "hommx_LinearElasticityHMM", **kw):
        super().__init__(msh, A, f, msh_micro, eps, petsc_options_global_solve, petsc_options_cell_problem,
                         petsc_options_prefix, **kw)
        self._Dtheta_t = Dtheta_transpose


class PoissonPeriodicHMM:
    """Classical periodic homogenisation, A = A(y) (hmm.py:1070-1279): one cell problem -> constant A_hom -> plain FEM
    macro solve.  ``compute_effective_tensor`` (hmm.py:1219-1245) is a batch of ONE on the GPU."""

    def __init__(self, msh, A, f, msh_micro, eps, petsc_options_global_solve=None, petsc_options_cell_problem=None,
                 petsc_options_prefix: str = "hommx_periodicHMM", **kw):
        self._inner = PoissonHMM(msh, lambda x, y: A(y), f, msh_micro, eps, petsc_options_global_solve,
                                 petsc_options_cell_problem, petsc_options_prefix, **kw)
        self._inner._bcs = []
        self._A_hom = None
        self._correctors = None

    @property
    def function_space(self):
        return self._inner.function_space

    @property
    def A_hom(self):
        return self._A_hom

    def set_boundary_conditions(self, bcs):
        self._inner.set_boundary_conditions(bcs)

    def set_right_hand_side(self, f):
        self._inner.set_right_hand_side(f)

    @property
    def correctors(self) -> list[fem.Function]:
        """One corrector per direction e_q on the micro mesh (hmm.py:1211-1213, filled by compute_effective_tensor :1239-1240)."""
        if self._correctors is None:
            self.compute_effective_tensor()
        return self._correctors

    def compute_effective_tensor(self) -> np.ndarray:
        """hmm.py:1219-1245: one periodic cell problem per direction -> correctors and A_hom (a batch of ONE on the GPU)."""
        h = self._inner
        cells = np.array([0])
        coef, kind = h._element_means(cells)
        AH, chi, info = h._ensure_plan(kind).solve(coef, None, return_info=True, return_correctors=True)
        if info[0]:
            h._logger.error("Something went wrong in the cell problem solving for the periodic cell")
        self._A_hom = AH[0]
        idx = h._periodic_to_micro_nodes()
        V_micro = fem.FunctionSpace(h._cell_mesh, 1)
        self._correctors = []
        for q in range(chi.shape[1]):
            f = fem.Function(V_micro)
            f.x.array[:] = chi[0, q][idx]
            self._correctors.append(f)
        return self._A_hom

    def solve(self) -> fem.Function:
        """hmm.py:1247-1256: standard P1 FEM with the constant A_hom."""
        if self._A_hom is None:
            self.compute_effective_tensor()
        h = self._inner
        cells = np.arange(h._msh.num_cells)
        S = h._local_stiffness_from_tensors(cells, np.broadcast_to(self._A_hom, (len(cells),) + self._A_hom.shape))
        dofs = h._msh.cells.astype(np.int64)
        nb = dofs.shape[1]
        N = h._num_global_dofs
        h._A = sp.coo_matrix((S.ravel(), (np.repeat(dofs, nb, axis=1).ravel(), np.tile(dofs, (1, nb)).ravel())),
                             shape=(N, N)).tocsr()
        h._needs_reassembly = False
        self._lp_A = h._A
        return h.solve()
