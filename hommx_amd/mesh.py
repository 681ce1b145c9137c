"""Light structured simplicial meshes standing in for ``dolfinx.mesh`` (DOLFINx is not a dependency).

The reference takes ``dolfinx.mesh.Mesh`` objects built by ``mesh.create_unit_square`` /
``create_rectangle`` / ``create_unit_cube`` / ``create_box`` (e.g. test_integration_poisson.py:76-83,
test_integration_linear_elasticity.py:32-49, 184-202; rotated_fibers.py:82-89).  The functions below keep
those names and argument order (minus the MPI communicator) and reproduce DOLFINx's default
triangulation: ``DiagonalType.right`` squares (v0,v1,v3),(v0,v2,v3) and six tetrahedra per cube
around the v0-v7 diagonal.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class Topology:
    dim: int


@dataclass
class Geometry:
    x: np.ndarray  # [n_vertices, 3]  (padded with zeros like dolfinx)
    dim: int


@dataclass
class Mesh:
    """P1 simplicial mesh.  ``geometry.x`` is [n_vertices, 3] as in DOLFINx; ``cells`` is [n_cells, dim+1]."""

    geometry: Geometry
    topology: Topology
    cells: np.ndarray
    shape: tuple = field(default_factory=tuple)  # (nx, ny[, nz]) when structured
    comm: object = None

    @property
    def num_cells(self) -> int:
        return self.cells.shape[0]

    @property
    def num_vertices(self) -> int:
        return self.geometry.x.shape[0]

    def cell_vertices(self) -> np.ndarray:
        """[n_cells, dim+1, dim] vertex coordinates per cell."""
        return self.geometry.x[self.cells][:, :, : self.topology.dim]

    def cell_midpoints(self) -> np.ndarray:
        """c_T = mean of the cell vertices (hmm.py:349-350), padded to 3 components like x_macro (hmm.py:190-192)."""
        return self.geometry.x[self.cells].mean(axis=1)

    def cell_volumes(self) -> np.ndarray:
        """hmm.py:20-28."""
        X = self.cell_vertices()
        J = X[:, 1:, :] - X[:, :1, :]
        return np.abs(np.linalg.det(J)) / (2.0 if self.topology.dim == 2 else 6.0)


def create_rectangle(points, n, comm=None) -> Mesh:
    (x0, y0), (x1, y1) = np.asarray(points[0], float)[:2], np.asarray(points[1], float)[:2]
    nx, ny = int(n[0]), int(n[1])
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    x = np.zeros(((nx + 1) * (ny + 1), 3))
    x[:, 0] = x0 + (x1 - x0) * ii.ravel() / nx
    x[:, 1] = y0 + (y1 - y0) * jj.ravel() / ny
    ci, cj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (cj * (nx + 1) + ci).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    cells = np.stack([np.stack([v0, v1, v3], 1), np.stack([v0, v2, v3], 1)], axis=1).reshape(-1, 3)
    return Mesh(Geometry(x, 2), Topology(2), cells.astype(np.int32), (nx, ny), comm)


def create_unit_square(nx: int, ny: int, comm=None) -> Mesh:
    return create_rectangle([(0.0, 0.0), (1.0, 1.0)], (nx, ny), comm)


def create_box(points, n, comm=None) -> Mesh:
    p0, p1 = np.asarray(points[0], float), np.asarray(points[1], float)
    nx, ny, nz = (int(v) for v in n)
    kk, jj, ii = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    x = np.stack(
        [
            p0[0] + (p1[0] - p0[0]) * ii.ravel() / nx,
            p0[1] + (p1[1] - p0[1]) * jj.ravel() / ny,
            p0[2] + (p1[2] - p0[2]) * kk.ravel() / nz,
        ],
        axis=1,
    )
    ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (ck * (ny + 1) * (nx + 1) + cj * (nx + 1) + ci).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    off = (nx + 1) * (ny + 1)
    v4, v5, v6, v7 = v0 + off, v1 + off, v2 + off, v3 + off
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.stack([np.stack(t, 1) for t in tets], axis=1).reshape(-1, 4)
    return Mesh(Geometry(x, 3), Topology(3), cells.astype(np.int32), (nx, ny, nz), comm)


def create_unit_cube(nx: int, ny: int, nz: int, comm=None) -> Mesh:
    return create_box([(0.0, 0.0, 0.0), (1.0, 1.0, 1.0)], (nx, ny, nz), comm)


def micro_cells_per_side(msh: Mesh) -> int:
    """n of a unit-cell mesh built by create_unit_square(n, n) / create_unit_cube(n, n, n)."""
    if not msh.shape or len(set(msh.shape)) != 1:
        raise ValueError("micro mesh must be a structured unit square/cube with equal cell counts per side")
    lo, hi = msh.geometry.x.min(axis=0), msh.geometry.x.max(axis=0)
    d = msh.topology.dim
    if not (np.allclose(lo[:d], 0.0) and np.allclose(hi[:d], 1.0)):
        raise ValueError("micro mesh needs to be the unit cell [0,1]^d (hmm.py:83)")
    return int(msh.shape[0])
