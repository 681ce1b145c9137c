"""Batched micro-cell solves: the Python face of the C ABI.

``MicroCellPlan.solve`` replaces the macro-cell loop of ``BaseHMM._assemble_stiffness``
(/root/reference/src/hommx/hmm.py:298-332): one call returns the effective tensor of every
macro cell of the batch.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

KINDS = {
    "poisson": _lib.KIND_POISSON_SCALAR,
    "poisson_matrix": _lib.KIND_POISSON_MATRIX,
    "elasticity": _lib.KIND_ELASTICITY_ISO,
    "elasticity_voigt": _lib.KIND_ELASTICITY_VOIGT,
}


class MicroCellPlan:
    """Everything batch-independent for one (dim, n_micro, kind): kernel choice + device scratch.

    Replaces the per-right-hand-side ``dolfinx_mpc.LinearProblem`` construction of hmm.py:420-425.
    """

    def __init__(self, dim: int, n_micro: int, kind: str = "poisson", device: int = 0, flags: int = 0):
        if kind not in KINDS:
            raise ValueError(f"unknown kind {kind!r}; expected one of {sorted(KINDS)}")
        self._lib = _lib.load()
        self.dim, self.n_micro, self.kind, self.device = int(dim), int(n_micro), kind, int(device)
        desc = _lib.PlanDesc(self.dim, self.n_micro, KINDS[kind], self.device, int(flags))
        h = C.c_void_p()
        _lib.check(self._lib.hommx_plan_create(C.byref(h), C.byref(desc)), "hommx_plan_create")
        self._h = h
        self.n_el = int(self._lib.hommx_plan_num_elements(h))
        self.n_comp = int(self._lib.hommx_plan_coef_components(h))
        self.t = int(self._lib.hommx_plan_tensor_size(h))
        self.kernel = self._lib.hommx_plan_kernel_name(h).decode()
        self.route_detail = self._lib.hommx_plan_route_detail(h).decode()  # what that route launches for this plan (reports)
        self.flops_per_solve = float(self._lib.hommx_plan_flops_per_solve(h))  # dense flops of the route, by its own model

    def reserve(self, n_cells: int):
        """Allocate the device workspace for batches of up to ``n_cells`` now (otherwise the first solve does it)."""
        _lib.check(self._lib.hommx_plan_reserve(self._h, int(n_cells)), "hommx_plan_reserve")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hommx_plan_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- host arrays -----------------------------------------------------------------------------
    def solve(self, coef: np.ndarray, M: np.ndarray | None = None, return_info: bool = False,
              return_correctors: bool = False):
        """coef[N_c, n_el(, n_comp)] float64, M[N_c, d, d] or None -> A_eff[N_c, t, t] (and info[N_c]).

        With ``return_correctors`` the result is (A_eff, correctors[N_c, t, n^d * bs][, info]): the periodic cell solutions
        of the canonical loads (mean-free), dof = node * bs + component."""
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        nc = coef.shape[0]
        if coef.size != nc * self.n_el * self.n_comp:
            raise ValueError(
                f"coef has shape {coef.shape}; expected ({nc}, {self.n_el}"
                + (f", {self.n_comp})" if self.n_comp > 1 else ")")
            )
        Mp = None
        if M is not None:
            M = np.ascontiguousarray(M, dtype=np.float64)
            if M.shape != (nc, self.dim, self.dim):
                raise ValueError(f"M has shape {M.shape}; expected ({nc}, {self.dim}, {self.dim})")
            Mp = M.ctypes.data
        out = np.empty((nc, self.t, self.t), dtype=np.float64)
        info = np.zeros(nc, dtype=np.int32)
        if return_correctors:
            bs = 1 if self.kind.startswith("poisson") else self.dim
            corr = np.empty((nc, self.t, self.n_micro**self.dim * bs), dtype=np.float64)
            if nc:
                _lib.check(
                    self._lib.hommx_solve_batch_correctors(
                        self._h, nc, coef.ctypes.data, Mp, out.ctypes.data, corr.ctypes.data, info.ctypes.data
                    ),
                    "hommx_solve_batch_correctors",
                )
            return (out, corr, info) if return_info else (out, corr)
        if nc:
            _lib.check(
                self._lib.hommx_solve_batch(self._h, nc, coef.ctypes.data, Mp, out.ctypes.data, info.ctypes.data),
                "hommx_solve_batch",
            )
        return (out, info) if return_info else out

    def solve_two_phase(self, mask: np.ndarray, values: np.ndarray, M: np.ndarray | None = None,
                        return_info: bool = False):
        """Two-phase media sampled on the device: mask[n_el] (bool / uint8, phase of every micro element) and
        values[N_c, 2(, n_comp)] = coefficient of phase 0 / phase 1 at every macro cell -> A_eff[N_c, t, t]."""
        mask = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if mask.shape != (self.n_el,):
            raise ValueError(f"mask has shape {mask.shape}; expected ({self.n_el},)")
        values = np.ascontiguousarray(values, dtype=np.float64)
        nc = values.shape[0]
        if values.size != nc * 2 * self.n_comp:
            raise ValueError(f"values has shape {values.shape}; expected ({nc}, 2" + (f", {self.n_comp})" if self.n_comp > 1 else ")"))
        Mp = None
        if M is not None:
            M = np.ascontiguousarray(M, dtype=np.float64)
            if M.shape != (nc, self.dim, self.dim):
                raise ValueError(f"M has shape {M.shape}; expected ({nc}, {self.dim}, {self.dim})")
            Mp = M.ctypes.data
        out = np.empty((nc, self.t, self.t), dtype=np.float64)
        info = np.zeros(nc, dtype=np.int32)
        if nc:
            _lib.check(
                self._lib.hommx_solve_batch_two_phase(
                    self._h, nc, mask.ctypes.data, values.ctypes.data, Mp, out.ctypes.data, info.ctypes.data
                ),
                "hommx_solve_batch_two_phase",
            )
        return (out, info) if return_info else out

    def solve_separable(self, family: str, table: np.ndarray, weights: np.ndarray | None, params: np.ndarray,
                        M: np.ndarray | None = None, return_info: bool = False):
        """Separable coefficient sampled on the device (include/hommx_hip.h): ``family`` "affine" (table[n_el] = element means of
        g) or "reciprocal" (table[n_el, n_q] = g at the quadrature points, weights[n_q]); params[N_c, 2] = (a, b) per macro cell."""
        fam = {"affine": _lib.SAMPLER_AFFINE, "reciprocal": _lib.SAMPLER_RECIPROCAL}[family]
        table = np.ascontiguousarray(table, dtype=np.float64)
        nq = 1 if fam == _lib.SAMPLER_AFFINE else int(table.shape[1])
        if table.size != self.n_el * nq:
            raise ValueError(f"table has shape {table.shape}; expected ({self.n_el}" + (f", {nq})" if nq > 1 else ",)"))
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if fam == _lib.SAMPLER_RECIPROCAL and (w is None or w.shape != (nq,)):
            raise ValueError(f"weights must have shape ({nq},)")
        params = np.ascontiguousarray(params, dtype=np.float64)
        nc = params.shape[0]
        want = (nc, 2) if self.n_comp == 1 else (nc, self.n_comp, 2)
        if params.shape != want:
            raise ValueError(f"params has shape {params.shape}; expected {want}")
        Mp = None
        if M is not None:
            M = np.ascontiguousarray(M, dtype=np.float64)
            if M.shape != (nc, self.dim, self.dim):
                raise ValueError(f"M has shape {M.shape}; expected ({nc}, {self.dim}, {self.dim})")
            Mp = M.ctypes.data
        out = np.empty((nc, self.t, self.t), dtype=np.float64)
        info = np.zeros(nc, dtype=np.int32)
        if nc:
            _lib.check(
                self._lib.hommx_solve_batch_separable(self._h, nc, fam, nq, table.ctypes.data, None if w is None else w.ctypes.data,
                                                      params.ctypes.data, Mp, out.ctypes.data, info.ctypes.data),
                "hommx_solve_batch_separable",
            )
        return (out, info) if return_info else out

    def solve_separable_device(self, n_cells: int, family: str, n_q: int, table_ptr: int, weights_ptr: int | None, params_ptr: int,
                               M_ptr: int | None, out_ptr: int, info_ptr: int | None, stream: int | None = None):
        fam = {"affine": _lib.SAMPLER_AFFINE, "reciprocal": _lib.SAMPLER_RECIPROCAL}[family]
        _lib.check(
            self._lib.hommx_solve_batch_separable_device(self._h, int(n_cells), fam, int(n_q), table_ptr, weights_ptr or None, params_ptr,
                                                         M_ptr or None, out_ptr, info_ptr or None, stream or None),
            "hommx_solve_batch_separable_device",
        )

    def solve_two_phase_device(self, n_cells: int, mask_ptr: int, values_ptr: int, M_ptr: int | None, out_ptr: int,
                               info_ptr: int | None, stream: int | None = None):
        _lib.check(
            self._lib.hommx_solve_batch_two_phase_device(
                self._h, int(n_cells), mask_ptr, values_ptr, M_ptr or None, out_ptr, info_ptr or None, stream or None
            ),
            "hommx_solve_batch_two_phase_device",
        )

    # -- device pointers (torch tensors or raw ints), asynchronous --------------------------------
    def solve_device(self, n_cells: int, coef_ptr: int, M_ptr: int | None, out_ptr: int, info_ptr: int | None,
                     stream: int | None = None):
        _lib.check(
            self._lib.hommx_solve_batch_device(
                self._h, int(n_cells), coef_ptr, M_ptr or None, out_ptr, info_ptr or None, stream or None
            ),
            "hommx_solve_batch_device",
        )
