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

if __name__ == "__main__":
    for name, args in CASES.items():
        d = case(*args)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind=args[0], dim=args[1], n=args[2], **d)
        print(name, d["A_eff"].shape)
