"""Minimal P1 finite-element types standing in for the slice of ``dolfinx.fem`` / ``dolfinx.mesh`` that HOMMX's
public API touches (function spaces, functions, Dirichlet conditions, boundary location).  CPU only: this is
the macro side, which stays on the host (BASELINE.json north_star; reference hmm.py:434-491).
"""

from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
from scipy.special import roots_jacobi, roots_legendre

from .mesh import Mesh


@dataclass
class FunctionSpace:
    """Lagrange-1 space; ``bs`` = block size (1 scalar, d vector).  Dof i of component k is ``i*bs + k``
    (blocked, node-major -- the unrolling of hmm.py:31-40)."""

    mesh: Mesh
    bs: int = 1

    @property
    def num_nodes(self) -> int:
        return self.mesh.num_vertices

    @property
    def num_dofs(self) -> int:
        return self.mesh.num_vertices * self.bs

    def tabulate_dof_coordinates(self) -> np.ndarray:
        return self.mesh.geometry.x

    def cell_dofs(self, cell: int) -> np.ndarray:
        return self.mesh.cells[cell]


def functionspace(msh: Mesh, element) -> FunctionSpace:
    """``fem.functionspace(msh, ("Lagrange", 1))`` / ``("Lagrange", 1, (d,))`` (hmm.py:639, 882)."""
    family, degree = element[0], element[1]
    if family not in ("Lagrange", "P", "CG") or degree != 1:
        raise ValueError("only Lagrange-1 spaces are supported (the reference uses nothing else)")
    bs = 1
    if len(element) > 2:
        shape = element[2]
        bs = int(shape[0]) if len(shape) else 1
    return FunctionSpace(msh, bs)


class Function:
    """``fem.Function``: values in ``.x.array`` (blocked layout)."""

    def __init__(self, V: FunctionSpace):
        self.function_space = V
        self._V = V
        self.x = SimpleNamespace(array=np.zeros(V.num_dofs))

    def copy(self) -> "Function":
        g = Function(self.function_space)
        g.x.array[:] = self.x.array
        return g

    def interpolate(self, f):
        X = self.function_space.mesh.geometry.x.T
        v = np.asarray(f(X), dtype=float)
        bs = self.function_space.bs
        if bs == 1:
            self.x.array[:] = np.broadcast_to(v.reshape(-1) if v.ndim else v, (self.function_space.num_nodes,))
        else:
            self.x.array[:] = np.broadcast_to(v.reshape(bs, -1), (bs, self.function_space.num_nodes)).T.ravel()

    def eval_at_nodes(self) -> np.ndarray:
        bs = self.function_space.bs
        return self.x.array if bs == 1 else self.x.array.reshape(-1, bs)


@dataclass
class DirichletBC:
    """``fem.dirichletbc(value, dofs, V)``: ``dofs`` are NODE indices (blocked dofs, as dolfinx returns them);
    ``g`` is a scalar, a length-bs vector or a Function (hmm.py:459-467)."""

    g: object
    dofs: np.ndarray
    V: FunctionSpace

    def unrolled(self):
        """(unrolled dof indices, values) -- hmm.py:454-467."""
        bs = self.V.bs
        nodes = np.asarray(self.dofs, dtype=np.int64)
        idx = (nodes[:, None] * bs + np.arange(bs)[None, :]).ravel()
        if isinstance(self.g, Function):
            val = self.g.x.array[idx]
        else:
            g = np.asarray(self.g, dtype=float)
            val = np.full(idx.shape, float(g)) if g.ndim == 0 else np.tile(g, nodes.shape[0])
        return idx, val


def dirichletbc(value, dofs, V: FunctionSpace) -> DirichletBC:
    return DirichletBC(value, np.asarray(dofs), V)


def locate_dofs_geometrical(V: FunctionSpace, marker) -> np.ndarray:
    return np.nonzero(np.asarray(marker(V.mesh.geometry.x.T), dtype=bool))[0]


def locate_entities_boundary(msh: Mesh, dim: int, marker) -> np.ndarray:
    """Boundary facets (as sorted vertex tuples, one row per facet) all of whose vertices satisfy ``marker``."""
    tdim = msh.topology.dim
    if dim != tdim - 1:
        raise ValueError("only facets (dim = tdim-1) are supported")
    cells = msh.cells
    nv = tdim + 1
    facets = np.concatenate([np.delete(cells, a, axis=1) for a in range(nv)], axis=0)
    facets = np.sort(facets, axis=1)
    uniq, counts = np.unique(facets, axis=0, return_counts=True)
    bnd = uniq[counts == 1]
    mk = np.asarray(marker(msh.geometry.x.T), dtype=bool)
    return bnd[np.all(mk[bnd], axis=1)]


def locate_dofs_topological(V: FunctionSpace, entity_dim: int, entities: np.ndarray) -> np.ndarray:
    return np.unique(np.asarray(entities).ravel())


# ---- quadrature on the reference simplex (macro right-hand side int f v) ------------------------------


def simplex_quadrature(dim: int, degree: int):
    """Collapsed Gauss-Jacobi rule exact for polynomials of total degree <= ``degree``:
    barycentric points [nq, dim+1] and weights summing to 1."""
    m = degree // 2 + 1
    if dim == 2:
        x0, w0 = roots_jacobi(m, 1.0, 0.0)
        x1, w1 = roots_legendre(m)
        a = 0.5 * (x0 + 1.0)[:, None]
        b = 0.5 * (x1 + 1.0)[None, :]
        l1 = a * np.ones_like(b)
        l2 = (1.0 - a) * b
        w = (w0[:, None] * w1[None, :]) * 0.125
        pts = np.stack([1.0 - l1 - l2, l1, l2], axis=-1).reshape(-1, 3)
        w = w.ravel()
    else:
        x0, w0 = roots_jacobi(m, 2.0, 0.0)
        x1, w1 = roots_jacobi(m, 1.0, 0.0)
        x2, w2 = roots_legendre(m)
        a = 0.5 * (x0 + 1.0)[:, None, None]
        b = 0.5 * (x1 + 1.0)[None, :, None]
        c = 0.5 * (x2 + 1.0)[None, None, :]
        l1 = a * np.ones_like(b) * np.ones_like(c)
        l2 = (1.0 - a) * b * np.ones_like(c)
        l3 = (1.0 - a) * (1.0 - b) * c
        w = (w0[:, None, None] * w1[None, :, None] * w2[None, None, :]) / 64.0
        pts = np.stack([1.0 - l1 - l2 - l3, l1, l2, l3], axis=-1).reshape(-1, 4)
        w = w.ravel()
    return pts, w / w.sum()


def assemble_load_vector(V: FunctionSpace, f, degree: int = 6) -> np.ndarray:
    """b_i = int f . phi_i  (hmm.py:129-133, 445-450).  ``f(x)`` is NumPy-vectorised: x[gdim(3), npts] ->
    scalar / (npts,) / (bs,) / (bs, npts)."""
    msh = V.mesh
    d, bs = msh.topology.dim, V.bs
    pts, w = simplex_quadrature(d, degree)
    X = msh.geometry.x[msh.cells]  # [nc, d+1, 3]
    xq = np.einsum("qa,cak->cqk", pts, X)  # [nc, nq, 3]
    nc, nq = xq.shape[:2]
    fv = np.asarray(f(xq.reshape(-1, 3).T), dtype=float)
    if bs == 1:
        fv = np.broadcast_to(fv.reshape(-1) if fv.ndim else fv, (nc * nq,)).reshape(nc, nq, 1)
    else:
        if fv.ndim <= 1:
            fv = np.broadcast_to(fv.reshape(bs, 1), (bs, nc * nq))
        fv = fv.reshape(bs, nc, nq).transpose(1, 2, 0)
    vol = msh.cell_volumes()
    be = np.einsum("c,q,qa,cqk->cak", vol, w, pts, fv)  # [nc, d+1, bs]
    idx = (msh.cells[:, :, None].astype(np.int64) * bs + np.arange(bs)[None, None, :])
    return np.bincount(idx.ravel(), weights=be.ravel(), minlength=V.num_dofs)


def l2_error_squared(u: Function, exact, degree: int = 8) -> float:
    """int (u_h - u)^2 dx  (the quantity test_integration_poisson.py:140-143 asserts on; squared, no sqrt)."""
    V = u.function_space
    msh = V.mesh
    d = msh.topology.dim
    pts, w = simplex_quadrature(d, degree)
    X = msh.geometry.x[msh.cells]
    xq = np.einsum("qa,cak->cqk", pts, X)
    uh = np.einsum("qa,ca->cq", pts, u.x.array[msh.cells])
    ue = np.asarray(exact(xq.reshape(-1, 3).T)).reshape(uh.shape)
    return float(np.einsum("c,q,cq->", msh.cell_volumes(), w, (uh - ue) ** 2))
