"""ctypes face of oracle/hommx_oracle_c.c (TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__, bench.py's cpu_baseline leg).

``effective_tensor_batch_c`` has the signature of ``hommx_oracle.effective_tensor_batch`` for the 2D scalar Poisson kind and
runs OpenMP over the macro cells."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libhommx_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.hommx_oracle_poisson2d.restype = C.c_int
        lib.hommx_oracle_poisson2d.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.hommx_oracle_generic.restype = C.c_int
        lib.hommx_oracle_generic.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def effective_tensor_batch_c(n: int, coef: np.ndarray, M: np.ndarray | None = None, threads: int = 0, return_info: bool = False):
    """coef[N_c, 2 n^2], M[N_c, 2, 2] or None -> A_H[N_c, 2, 2] (NaN where a pivot failed); ``threads`` 0 = OpenMP default."""
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    nc = coef.shape[0]
    if coef.shape != (nc, 2 * n * n):
        raise ValueError(f"coef has shape {coef.shape}; expected ({nc}, {2 * n * n})")
    Mp = None
    if M is not None:
        M = np.ascontiguousarray(M, dtype=np.float64)
        Mp = M.ctypes.data
    out = np.empty((nc, 2, 2))
    info = np.zeros(nc, dtype=np.int32)
    used = load().hommx_oracle_poisson2d(n, nc, coef.ctypes.data, Mp, out.ctypes.data, info.ctypes.data, int(threads))
    if used < 0:
        raise ValueError("hommx_oracle_poisson2d: bad arguments (3 <= n <= 64)")
    effective_tensor_batch_c.threads_used = used
    return (out, info) if return_info else out


def effective_tensor_generic_c(kind: str, dim: int, n: int, coef: np.ndarray, M: np.ndarray | None = None) -> np.ndarray:
    """ONE cell through the element-by-element C restatement (hommx_oracle_generic): ``kind`` / ``coef`` as hommx_oracle.build_cell_problem
    takes them -- 'poisson' with coef[n_el] or [n_el, d, d]; 'elasticity' with coef[n_el, 2] = (lambda, mu) or [n_el, d, d, d, d]."""
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    if kind == "poisson":
        k = 0 if coef.ndim == 1 else 1
        t = dim
    else:
        k = 2 if coef.ndim == 2 else 3
        t = dim * (dim + 1) // 2
    Mp = None
    if M is not None:
        M = np.ascontiguousarray(M, dtype=np.float64)
        Mp = M.ctypes.data
    out = np.empty((t, t))
    rc = load().hommx_oracle_generic(dim, n, k, coef.ctypes.data, Mp, out.ctypes.data)
    if rc:
        raise ValueError("hommx_oracle_generic: bad arguments or a non-positive pivot")
    return out
