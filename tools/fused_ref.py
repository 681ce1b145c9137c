"""Matrix-level NumPy model of the fused 2D kernel's recurrences (dev tool; mirrors hommx_amd/csrc/fused2d.hip step by step).

Conventions of the kernel: padding FIRST (node column c lives at matrix index c + p0, p0 = NB - n, identity on the padding),
T = -S carried instead of S, conv-3 Gauss-Jordan exchange giving N' = T^-1 = -S^-1.  `run(coef, M, n, NB)` returns A_H and a
dict of the intermediates per elimination step, which tools/debug_fused.py compares with the kernel's debug dump.
"""
import numpy as np


def sweep_conv3(T):
    """In-place exchange sweep: returns T^-1 (row <- -u/d, column <- col/d, (K,K) <- 1/d, rest += col * (-u/d))."""
    T = T.copy()
    nb = T.shape[0]
    for K in range(nb):
        d = T[K, K]
        pinv = 1.0 / d
        u = T[K, :].copy()
        w = -u * pinv
        wt = w.copy()
        wt[K] = pinv - 1.0
        col = T[:, K].copy()
        T += np.outer(col, wt)
        T[K, :] = w
        T[K, K] = pinv
    return T


def stencil_rows(coef, M, n):
    """Per node row j: diag dg[j][c], E-coupling within the row ce[j][c] (c <-> c+1), coupling to the next row cN[j][c] (node (c,j+1)
    <- (c,j)) and cNE[j][c] ((c+1,j+1) <- (c,j)), loads p0[j][c], p1[j][c]; C0 sum.  coef[2 n^2] in element order 2 (i + n j) + s."""
    a = coef.reshape(n, n, 2)  # [j][i][s]
    m00, m01, m10, m11 = (1.0, 0.0, 0.0, 1.0) if M is None else M.ravel()
    al = 0.5 * (m00 * m00 + m10 * m10)
    be = 0.5 * (m01 * m01 + m11 * m11)
    ga = 0.5 * (m00 * m01 + m10 * m11)
    ab = al - 2.0 * ga + be
    dg = np.zeros((n, n)); ce = np.zeros((n, n)); cN = np.zeros((n, n)); cNE = np.zeros((n, n))
    p0 = np.zeros((n, n)); p1 = np.zeros((n, n))
    for j in range(n):
        cur, prev = a[j], a[(j - 1) % n]
        a0, a1 = cur[:, 0], cur[:, 1]
        a0m, a1m = np.roll(a0, 1), np.roll(a1, 1)
        b0, b1 = prev[:, 0], prev[:, 1]
        b0m, b1m = np.roll(b0, 1), np.roll(b1, 1)
        dg[j] = a0 * al + a1 * be + a0m * ab + b0m * be + b1m * al + b1 * ab
        ce[j] = (a0 + b1) * (ga - al)
        cN[j] = (a1 + a0m) * (ga - be)
        cNE[j] = -ga * (a0 + a1)
        p0[j] = a0 - a0m - b1m + b1
        p1[j] = a1 + a0m - b0m - b1
    return dg, ce, cN, cNE, p0, p1, a.sum(), (m00, m01, m10, m11)


def pad_vec(v, NB, fill=0.0):
    out = np.full(NB, fill)
    out[NB - len(v):] = v
    return out


def band_D(dg, ce, NB):
    """cyclic tridiagonal on the real indices, identity on the padding"""
    n = len(dg)
    p = NB - n
    D = np.eye(NB)
    for c in range(n):
        D[p + c, p + c] = dg[c]
        cp = (c + 1) % n
        D[p + c, p + cp] += ce[c] if cp != c else 0.0
        D[p + cp, p + c] = D[p + c, p + cp]
    return D


def band_E(cN, cNE, NB):
    """E[r][r] = cN[r], E[r][r-1] = cNE[r-1] (cyclic), zero on the padding"""
    n = len(cN)
    p = NB - n
    E = np.zeros((NB, NB))
    for c in range(n):
        E[p + c, p + c] = cN[c]
        E[p + c, p + (c - 1) % n] += cNE[(c - 1) % n]
    return E


def run(coef, M, n, NB, trace=False):
    dg, ce, cN, cNE, p0, p1, asum, (m00, m01, m10, m11) = stencil_rows(coef, M, n)
    tr = []
    T = -band_D(dg[0], ce[0], NB)
    Slast = band_D(dg[n - 1], ce[n - 1], NB)
    # W_0 = K[(., n-1), (., 0)] = U_{n-1} = E_{n-1}^T  (coupling of row 0 to row n-1 going "up" from n-1)
    W = band_E(cN[n - 1], cNE[n - 1], NB).T.copy()
    R = np.stack([pad_vec(p0[0], NB), pad_vec(p1[0], NB)])
    Rl = np.stack([pad_vec(p0[n - 1], NB), pad_vec(p1[n - 1], NB)])
    G = np.zeros((2, 2))
    for j in range(n - 1):
        E = band_E(cN[j], cNE[j], NB)
        if j == n - 2:
            W = W + E
        Np = sweep_conv3(T)
        Vp = W @ Np
        Slast = Slast + Vp @ W.T
        Vr = R @ Np
        G += Vr @ R.T
        Rl = Rl + Vr @ W.T
        if trace:
            tr.append(dict(T=T.copy(), Np=Np.copy(), W=W.copy(), Vp=Vp.copy(), Slast=Slast.copy(), Vr=Vr.copy(), R=R.copy(), Rl=Rl.copy()))
        if j < n - 2:
            W = Vp @ E.T
            T = -band_D(dg[j + 1], ce[j + 1], NB) - E @ Np @ E.T
            R = np.stack([pad_vec(p0[j + 1], NB), pad_vec(p1[j + 1], NB)]) + Vr @ E.T
    Tl = -Slast
    Tl[NB - 1, :] = 0.0
    Tl[:, NB - 1] = 0.0
    Tl[NB - 1, NB - 1] = -1.0
    Rl[:, NB - 1] = 0.0
    Np = sweep_conv3(Tl)
    Vr = Rl @ Np
    G += Vr @ Rl.T
    h = 1.0 / n
    sc = 0.25 * h * h
    c0 = 0.5 * h * h * asum
    Mm = np.array([[m00, m01], [m10, m11]])
    AH = c0 * np.eye(2) + sc * (Mm @ G @ Mm.T)
    return (AH, tr) if trace else AH


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import hommx_oracle as O
    rng = np.random.default_rng(0)
    for n, NB in ((32, 32), (20, 32), (17, 32), (16, 16), (5, 16), (3, 16), (3, 32)):
        coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(2, 2 * n * n)))
        M = np.eye(2)[None] + 0.35 * rng.standard_normal((2, 2, 2))
        for MM in (None, M):
            ref = O.effective_tensor_batch("poisson", 2, n, coef, MM)
            got = np.stack([run(coef[k], None if MM is None else MM[k], n, NB) for k in range(2)])
            print(n, NB, MM is not None, np.abs(got - ref).max() / np.abs(ref).max())
