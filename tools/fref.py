"""Reference flop count F_ref of one micro-cell solve: flops of a sparse Cholesky factorisation of the periodic
P1 stiffness matrix under a fill-reducing ordering, from a symbolic analysis of its sparsity pattern -- "so that a smarter
ordering is not penalised" when the roofline fraction is quoted next to the dense block-cyclic model the kernels execute.

    F_ref = sum_j c_j^2  +  4 nnz(L) nrhs          c_j = entries of column j of L (diagonal included)

(first term: factorisation, second: forward + backward substitution of the nrhs canonical loads).  The pattern is the
7-point (2D, right-diagonal triangles) / 15-point (3D, six tetrahedra around the main diagonal) node stencil on the n^d torus,
times bs x bs, with the last node's unknowns removed (gauge).  Orderings tried: geometric nested dissection of the torus (two
cuts per periodic direction) and SuperLU's minimum degree on A + A^T; the smaller count is F_ref.  SuperLU (scipy) is used
only as a symbolic tool here: SymmetricMode, no pivoting, on an SPD surrogate with the same pattern.

    python tools/fref.py            # prints / refreshes profiles/fref.json (3D n = 16 takes about a minute)
"""
import itertools
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def torus_pattern(dim, n, bs):
    if dim == 2:
        offs = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)]
    else:
        offs = {(0, 0, 0)}
        for d in itertools.product((0, 1), repeat=3):
            if any(d):
                offs.add(d)
                offs.add(tuple(-x for x in d))
        offs = sorted(offs)
    idx = np.arange(n**dim).reshape((n,) * dim)
    rows, cols = [], []
    for o in offs:
        nb = idx
        for ax, s in enumerate(o):
            nb = np.roll(nb, -s, axis=ax)
        rows.append(idx.ravel())
        cols.append(nb.ravel())
    r, c = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n**dim, n**dim)).tocsr()
    A.data[:] = 1.0
    return sp.kron(A, np.ones((bs, bs))).tocsr() if bs > 1 else A


def nd_order(dim, n):
    """Geometric nested dissection of the n^dim torus (separators last)."""
    coords = np.stack(np.meshgrid(*[np.arange(n)] * dim, indexing="ij"), -1).reshape(-1, dim)
    out = []

    def rec(sel, lo, hi, periodic):
        size = [hi[a] - lo[a] for a in range(dim)]
        if len(sel) <= 8 or max(size) <= 2:
            out.append(sel)
            return
        ax = int(np.argmax(size))
        c = coords[sel, ax]
        mid = lo[ax] + size[ax] // 2
        l1, h1, l2, h2 = list(lo), list(hi), list(lo), list(hi)
        if periodic[ax]:  # a ring needs two cuts
            sep, a, b = sel[(c == lo[ax]) | (c == mid)], sel[(c > lo[ax]) & (c < mid)], sel[c > mid]
            periodic = list(periodic)
            periodic[ax] = False
            l1[ax], h1[ax], l2[ax] = lo[ax] + 1, mid, mid + 1
        else:
            sep, a, b = sel[c == mid], sel[c < mid], sel[c > mid]
            h1[ax], l2[ax] = mid, mid + 1
        rec(a, l1, h1, periodic)
        rec(b, l2, h2, periodic)
        out.append(sep)

    rec(np.arange(n**dim), [0] * dim, [n] * dim, [True] * dim)
    return np.concatenate(out)


def symbolic_cholesky(A, perm=None, spec="NATURAL"):
    """(sum_j c_j^2, nnz(L)) of the Cholesky factor of a matrix with A's pattern."""
    if perm is not None:
        A = A[perm][:, perm]
    A = A.tocsc().astype(float)
    A.data[:] = -1.0
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsc()
    lu = spla.splu(A, permc_spec=spec, diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    cc = np.diff(lu.L.tocsc().indptr).astype(float)
    return float((cc**2).sum()), int(cc.sum())


def fref(dim, n, bs, nrhs):
    A = torus_pattern(dim, n, bs)
    nd = A.shape[0] - bs
    Ak = A[:nd][:, :nd]
    p = nd_order(dim, n)
    pk = (np.repeat(p * bs, bs) + np.tile(np.arange(bs), len(p))) if bs > 1 else p
    pk = pk[pk < nd]
    cand = {"nested_dissection": symbolic_cholesky(Ak, pk), "minimum_degree": symbolic_cholesky(Ak, None, "MMD_AT_PLUS_A")}
    best = min(cand, key=lambda k: cand[k][0] + 4.0 * cand[k][1] * nrhs)
    f, nnzL = cand[best]
    return {"dim": dim, "n": n, "bs": bs, "nrhs": nrhs, "unknowns": int(nd), "ordering": best, "nnz_L": nnzL,
            "factor_flops": f, "solve_flops": 4.0 * nnzL * nrhs, "F_ref": f + 4.0 * nnzL * nrhs,
            "candidates": {k: {"factor_flops": v[0], "nnz_L": v[1]} for k, v in cand.items()}}


CASES = {  # BASELINE.json configurations + the sizes of the reference's own tests
    "C1_poisson2d_n16": (2, 16, 1, 2),
    "C2_C3_poisson2d_n32": (2, 32, 1, 2),
    "C4_C5_elasticity3d_n16": (3, 16, 3, 6),
    "elasticity2d_n10": (2, 10, 2, 3),
    "poisson3d_n6": (3, 6, 1, 3),
}

if __name__ == "__main__":
    out = {}
    for name, args in CASES.items():
        t = time.time()
        out[name] = fref(*args)
        print(name, f"F_ref = {out[name]['F_ref']:.4e} ({out[name]['ordering']}, {time.time() - t:.1f} s)", flush=True)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "fref.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)
