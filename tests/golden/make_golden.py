"""Generate the golden vectors under tests/golden/ with the CPU oracle (run in the build container).

The reference ships no fixture files and cannot be imported here, so these vectors
are oracle outputs; the oracle itself is pinned by tests/test_oracle_kat.py against the reference
tests' analytic answers.  Inputs are seeded; re-running reproduces the files bit for bit.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import hommx_oracle as O  # noqa: E402


def case(kind, dim, n, ncell, seed, stratified):
    rng = np.random.default_rng(seed)
    n_el = (2 if dim == 2 else 6) * n**dim
    if kind == "poisson":
        coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(ncell, n_el)))
    else:
        coef = np.stack([rng.uniform(0.5, 2.0, (ncell, n_el)), np.exp(rng.uniform(np.log(0.01), np.log(100.0), (ncell, n_el)))], axis=-1)
    M = np.eye(dim)[None] + 0.35 * rng.standard_normal((ncell, dim, dim)) if stratified else None
    AH = O.effective_tensor_batch(kind, dim, n, coef, M)
    return dict(coef=coef, M=np.zeros(0) if M is None else M, A_eff=AH)


CASES = {
    "poisson2d_n8": ("poisson", 2, 8, 4, 1, False),
    "poisson2d_n15_strat": ("poisson", 2, 15, 4, 2, True),
    "poisson2d_n16": ("poisson", 2, 16, 3, 3, False),
    "poisson2d_n32_strat": ("poisson", 2, 32, 3, 4, True),
    "poisson3d_n4": ("poisson", 3, 4, 3, 5, False),
    "poisson3d_n6_strat": ("poisson", 3, 6, 2, 6, True),
    "elasticity2d_n10": ("elasticity", 2, 10, 3, 7, False),
    "elasticity2d_n6_strat": ("elasticity", 2, 6, 3, 8, True),
    "elasticity3d_n3": ("elasticity", 3, 3, 3, 9, False),
    "elasticity3d_n4_strat": ("elasticity", 3, 4, 2, 10, True),
}


def fibre_cells(which):
    """Production-size cells (16^3 micro cells, 12288 unknowns) of BASELINE configs C4 / C5 (coefficient of
    examples/linear_elasticity/rotated_fibers.py:23-76, forms hmm.py:887-922 / 1024-1067).  The fibre coefficient is two-phase, so
    the fixture stores the packed phase mask (one bit per tet), the two (lambda, mu) phase values of every cell, M and the
    oracle's C_H; `expand_fibre_fixture` below rebuilds coef[cell][n_el][2] from them.  About 25 s of oracle time per cell."""
    from hommx_amd import workloads as W

    n = 16
    if which == "c4":
        msh, coef, M = W.c4_fibre_beam()
        c = msh.cell_midpoints()
        # lowest / middle / highest contrast mu_in(x0) = 100 (1 + x0) against mu_out = 0.001
        order = np.argsort(c[:, 0], kind="stable")
        cells = np.array([order[0], order[len(order) // 2], order[-1]])
    else:
        msh, coef, M = W.c5_rotated_fibres()
        asym = np.abs(M - np.transpose(M, (0, 2, 1))).max(axis=(1, 2))
        mid = np.abs(M[:, 0, 2] * M[:, 2, 2])  # |sin g cos g|: the fibre rotated by about 45 degrees
        # most non-symmetric M (largest d theta_2 / d x_1), a half-rotated fibre, the least non-symmetric M
        cells = np.array([int(np.argmax(asym)), int(np.argmax(mid)), int(np.argmin(asym))])
        assert len(set(cells.tolist())) == 3
    inside = coef[0, :, 1] != 0.001
    for k in cells:
        assert np.array_equal(coef[k, :, 1] != 0.001, inside)
    values = np.stack([np.stack([coef[k][~inside][0], coef[k][inside][0]]) for k in cells])  # [cell][phase][lam, mu]
    Msel = None if M is None else M[cells]
    AH = O.effective_tensor_batch("elasticity", 3, n, coef[cells], Msel)
    return dict(cells=cells, mask_bits=np.packbits(inside), values=values, M=np.zeros(0) if Msel is None else Msel, A_eff=AH,
                midpoints=msh.cell_midpoints()[cells])


def expand_fibre_fixture(g):
    """coef[cell][n_el][2] of a c4_n16 / c5_n16_strat fixture."""
    n_el = 6 * int(g["n"]) ** 3
    inside = np.unpackbits(g["mask_bits"])[:n_el].astype(bool)
    return np.where(inside[None, :, None], g["values"][:, 1][:, None, :], g["values"][:, 0][:, None, :])


FULL_SIZE = {"c4_n16": "c4", "c5_n16_strat": "c5"}

if __name__ == "__main__":
    only = sys.argv[1:]
    for name, args in CASES.items():
        if only and name not in only:
            continue
        d = case(*args)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind=args[0], dim=args[1], n=args[2], **d)
        print(name, d["A_eff"].shape)
    for name, which in FULL_SIZE.items():
        if only and name not in only:
            continue
        d = fibre_cells(which)
        np.savez_compressed(os.path.join(HERE, "fullsize_" + name + ".npz"), kind="elasticity", dim=3, n=16, **d)
        print(name, d["cells"], d["A_eff"].shape)
