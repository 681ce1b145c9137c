"""test/unit/test_unit.py of the reference, on hommx_amd.cell_problem (CPU): geometry of the periodic master/slave maps."""

import numpy as np

from hommx_amd import fem, mesh
from hommx_amd.cell_problem import create_periodic_boundary_conditions


def test_periodic_boundary_conditions_unit_square():
    """test_unit.py:25-54."""
    N = 7
    msh = mesh.create_unit_square(N, N)
    V = fem.functionspace(msh, ("Lagrange", 1))
    mpc = create_periodic_boundary_conditions(V)
    x = V.tabulate_dof_coordinates()
    # only dofs on the (max) boundary are slaves
    assert np.all(np.isclose(x[mpc.slaves, 0], 1) | np.isclose(x[mpc.slaves, 1], 1))
    # (1,1) is mapped to (0,0)
    corner = mpc.slaves[np.isclose(x[mpc.slaves, 0], 1) & np.isclose(x[mpc.slaves, 1], 1)]
    assert len(corner) == 1
    assert np.allclose(x[mpc.masters[mpc.slaves == corner[0]][0]], 0)
    # every pair differs by unit vectors only, masters are never slaves
    diff = (x[mpc.slaves] - x[mpc.masters])[:, :2]
    assert np.all(np.isclose(diff, 0) | np.isclose(diff, 1))
    assert np.all(np.isclose(diff, 1).sum(axis=1) >= 1)
    assert not set(mpc.masters) & set(mpc.slaves)
    assert len(mpc.slaves) == 2 * N + 1 and mpc.num_independent == N * N
    assert len(np.unique(mpc.to_periodic)) == N * N
    assert np.array_equal(mpc.to_periodic[mpc.slaves], mpc.to_periodic[mpc.masters])


def test_periodic_boundary_conditions_unit_cube():
    """test_unit.py:57-103: faces -> opposite faces, edges -> opposite edges, corner -> origin."""
    N = 4
    msh = mesh.create_unit_cube(N, N, N)
    V = fem.functionspace(msh, ("Lagrange", 1))
    mpc = create_periodic_boundary_conditions(V)
    x = V.tabulate_dof_coordinates()
    on_max = np.isclose(x[mpc.slaves], 1)
    assert np.all(on_max.any(axis=1))
    assert len(mpc.slaves) == (N + 1) ** 3 - N**3
    diff = x[mpc.slaves] - x[mpc.masters]
    # the displacement is exactly the set of max-coordinates of the slave
    assert np.allclose(diff, on_max.astype(float))
    # the corner (1,1,1) goes to the origin; an edge slave (1,1,z) goes to (0,0,z)
    c = np.nonzero(on_max.all(axis=1))[0]
    assert len(c) == 1 and np.allclose(x[mpc.masters[c[0]]], 0)
    e = np.nonzero(on_max[:, 0] & on_max[:, 1] & ~on_max[:, 2])[0]
    assert np.allclose(x[mpc.masters[e]][:, :2], 0) and np.allclose(x[mpc.masters[e]][:, 2], x[mpc.slaves[e]][:, 2])
    assert not set(mpc.masters) & set(mpc.slaves)
    # vector-valued space: same node map, block size 3
    Vv = fem.functionspace(msh, ("Lagrange", 1, (3,)))
    mv = create_periodic_boundary_conditions(Vv)
    vals = np.arange(mv.num_independent * 3, dtype=float)
    full = mv.backsubstitution(vals).reshape(-1, 3)
    assert np.array_equal(full[mv.slaves], full[mv.masters])
