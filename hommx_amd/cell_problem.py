"""Periodic cell problems on the unit square / cube -- the counterpart of ``hommx.cell_problem``
(/root/reference/src/hommx/cell_problem.py).

The reference builds the periodic identification with ``dolfinx_mpc`` (slave dofs on the max faces, masters on the
min faces; doubly / triply constrained edges and the corner handled separately so that no slave is a master:
cell_problem.py:38-136 in 2D, :139-300 in 3D) and re-creates a ``PeriodicLinearProblem`` (MPC assembly + null space +
KSP solve + back substitution, cell_problem.py:303-388) for every right-hand side of every macro cell.  On the structured
unit-cell mesh the net effect of the constraint is the torus map ``(i, j[, k]) -> (i mod n, j mod n[, k mod n])``; here it
is an index table built once per micro-mesh size, and the solve is one batched GPU call (``hommx_amd.batch``).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import fem
from .batch import MicroCellPlan
from .mesh import Mesh, micro_cells_per_side


@dataclass
class PeriodicConstraint:
    """What ``dolfinx_mpc.MultiPointConstraint`` holds after ``finalize()`` for the periodic unit cell.

    ``slaves`` / ``masters`` are node indices of the micro mesh (blocked: dof = node * bs + component);
    ``to_periodic[node]`` is the index of the independent unknown the node is identified with, ``num_independent`` = n^d.
    """

    function_space: fem.FunctionSpace
    slaves: np.ndarray
    masters: np.ndarray
    to_periodic: np.ndarray
    num_independent: int

    def backsubstitution(self, periodic_values: np.ndarray) -> np.ndarray:
        """Values at the independent unknowns -> values at all mesh nodes (``mpc.backsubstitution``, cell_problem.py:386)."""
        bs = self.function_space.bs
        v = np.asarray(periodic_values).reshape(self.num_independent, bs)
        return v[self.to_periodic].ravel()


def create_periodic_boundary_conditions(function_space: fem.FunctionSpace, bcs=None) -> PeriodicConstraint:
    """Periodic boundary condition on the unit square or unit cube (cell_problem.py:16-35).

    Every node on a max face is a slave of its image on the min faces: faces first (cell_problem.py:113-120, 237-245), then
    the doubly constrained edges (:248-295) and the (triply) constrained corner (:123-134, :276-298).  ``bcs`` is accepted
    for signature compatibility and ignored, as in the reference.
    """
    msh: Mesh = function_space.mesh
    d = msh.topology.dim
    if d == 1:
        raise ValueError("Periodic boundary conditions in 1d not implemented.")  # cell_problem.py:27-28
    if d not in (2, 3):
        raise ValueError(f"Unkown topology dimension. {d=} is something unexpected")
    n = micro_cells_per_side(msh)
    x = msh.geometry.x[:, :d]
    lo, hi = x.min(axis=0), x.max(axis=0)
    on_max = np.isclose(x, hi)
    slaves = np.nonzero(on_max.any(axis=1))[0]
    g = np.rint((x - lo) / (hi - lo) * n).astype(np.int64)
    gm = g[slaves].copy()
    gm[on_max[slaves]] = 0
    stride = (n + 1) ** np.arange(d)
    masters = gm @ stride
    to_periodic = (g % n) @ (n ** np.arange(d))
    return PeriodicConstraint(function_space, slaves, masters, to_periodic, n**d)


class PeriodicLinearProblem:
    """One periodic cell problem ``a(chi, z) = -l_m(z)`` for all canonical loads m at once -- the role of
    ``PeriodicLinearProblem`` (cell_problem.py:303-388) for the bilinear forms of ``hommx.hmm``.

    The reference takes UFL forms ``a`` and ``L``; without UFL the problem is described by its kind and data:

        kind   'poisson' | 'poisson_matrix' | 'elasticity' | 'elasticity_voigt'      (hmm.py:644-650 / 891-903)
        coef   element means of the coefficient on the micro mesh, in mesh element order
        M      optional Dtheta^T (stratified forms, hmm.py:759-772 / 1032-1048)

    ``solve()`` returns one ``fem.Function`` per canonical load on the micro mesh (slave nodes filled by back substitution,
    constants projected out as the reference's null-space handling does, cell_problem.py:349-361, 382) and stores the
    effective tensor in ``effective_tensor``.  ``petsc_options`` is accepted and ignored (direct factorisation).
    """

    def __init__(self, kind: str, coef: np.ndarray, mpc: PeriodicConstraint, M: np.ndarray | None = None,
                 petsc_options: dict | None = None, device: int = 0):
        self._mpc = mpc
        self._kind = kind
        msh = mpc.function_space.mesh
        self._dim = msh.topology.dim
        self._n = micro_cells_per_side(msh)
        self._coef = np.asarray(coef, dtype=float)[None]
        self._M = None if M is None else np.asarray(M, dtype=float)[None]
        self._plan = MicroCellPlan(self._dim, self._n, kind, device=device)
        self.effective_tensor: np.ndarray | None = None
        self.info: int | None = None

    def solve(self) -> list[fem.Function]:
        AH, chi, info = self._plan.solve(self._coef, self._M, return_info=True, return_correctors=True)
        self.effective_tensor, self.info = AH[0], int(info[0])
        out = []
        for m in range(chi.shape[1]):
            f = fem.Function(self._mpc.function_space)
            f.x.array[:] = self._mpc.backsubstitution(chi[0, m])
            out.append(f)
        return out
