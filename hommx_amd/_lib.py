"""ctypes binding of libhommx_hip.so (include/hommx_hip.h).

The library is the product path; there is NO CPU fallback.  If the shared object is missing or
cannot be loaded, every entry point raises -- loudly -- instead of computing something else.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOMMX_LIB") or os.path.join(_HERE, "libhommx_hip.so")  # HOMMX_LIB: A/B builds (dev)

# every symbol include/hommx_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "hommx_device_count",
    "hommx_plan_create",
    "hommx_plan_destroy",
    "hommx_plan_reserve",
    "hommx_plan_dim",
    "hommx_plan_device",
    "hommx_plan_n_micro",
    "hommx_plan_kind",
    "hommx_plan_num_elements",
    "hommx_plan_coef_components",
    "hommx_plan_tensor_size",
    "hommx_plan_kernel_name",
    "hommx_plan_route_detail",
    "hommx_plan_flops_per_solve",
    "hommx_solve_batch",
    "hommx_solve_batch_device",
    "hommx_solve_batch_correctors",
    "hommx_solve_batch_two_phase",
    "hommx_solve_batch_two_phase_device",
    "hommx_solve_batch_separable",
    "hommx_solve_batch_separable_device",
    "hommx_comm_init_all",
    "hommx_comm_destroy",
    "hommx_comm_size",
    "hommx_allgather_field",
    "hommx_solve_batch_multi",
    "hommx_solve_batch_multi_device",
    "hommx_shard_range",
    "hommx_unpack_field",
    "hommx_calibrate_fp64",
    "hommx_calibrate_fp64_mfma",
    "hommx_calibrate_fp64_detail",
    "hommx_last_error",
)

KIND_POISSON_SCALAR = 0
KIND_POISSON_MATRIX = 1
KIND_ELASTICITY_ISO = 2
KIND_ELASTICITY_VOIGT = 3
SAMPLER_AFFINE = 0
SAMPLER_RECIPROCAL = 1


class PlanDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int32),
        ("n_micro", C.c_int32),
        ("kind", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class HommxLibraryError(RuntimeError):
    pass


_lib = None


def _share_hip_runtime_with_torch():
    """One process, one HIP runtime.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so.7``; libhommx_hip.so needs the
    same SONAME.  Whichever copy is loaded first serves both, and torch cannot find a GPU when that is not its own copy
    ("No HIP GPUs are available" if libhommx_hip.so was loaded before ``import torch``).  So when torch is installed but not
    imported yet, its copy is loaded first, by path (no ``import torch``: the product does not depend on it)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:  # pragma: no cover - fall back to the system runtime
            pass


def load():
    """Load libhommx_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HommxLibraryError(
            f"{LIB_PATH} not found: build it with `make -C hommx_amd/csrc` (or __graft_entry__.build()). "
            "hommx_amd has no CPU fallback for the micro-cell solves."
        )
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise HommxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
    lib.hommx_device_count.restype = C.c_int
    lib.hommx_device_count.argtypes = []
    lib.hommx_plan_create.restype = C.c_int
    lib.hommx_plan_create.argtypes = [C.POINTER(vp), C.POINTER(PlanDesc)]
    lib.hommx_plan_reserve.restype = C.c_int
    lib.hommx_plan_reserve.argtypes = [vp, i64]
    lib.hommx_plan_destroy.restype = C.c_int
    lib.hommx_plan_destroy.argtypes = [vp]
    lib.hommx_plan_num_elements.restype = i64
    lib.hommx_plan_num_elements.argtypes = [vp]
    lib.hommx_plan_coef_components.restype = i32
    lib.hommx_plan_coef_components.argtypes = [vp]
    lib.hommx_plan_tensor_size.restype = i32
    lib.hommx_plan_tensor_size.argtypes = [vp]
    lib.hommx_plan_flops_per_solve.restype = C.c_double
    lib.hommx_plan_flops_per_solve.argtypes = [vp]
    lib.hommx_plan_kernel_name.restype = C.c_char_p
    lib.hommx_plan_kernel_name.argtypes = [vp]
    lib.hommx_plan_route_detail.restype = C.c_char_p
    lib.hommx_plan_route_detail.argtypes = [vp]
    lib.hommx_solve_batch.restype = C.c_int
    lib.hommx_solve_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.hommx_solve_batch_device.restype = C.c_int
    lib.hommx_solve_batch_device.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.hommx_solve_batch_correctors.restype = C.c_int
    lib.hommx_solve_batch_correctors.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.hommx_solve_batch_two_phase.restype = C.c_int
    lib.hommx_solve_batch_two_phase.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.hommx_solve_batch_two_phase_device.restype = C.c_int
    lib.hommx_solve_batch_two_phase_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.hommx_solve_batch_separable.restype = C.c_int
    lib.hommx_solve_batch_separable.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.hommx_solve_batch_separable_device.restype = C.c_int
    lib.hommx_solve_batch_separable_device.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    for q in ("hommx_plan_dim", "hommx_plan_device", "hommx_plan_n_micro", "hommx_plan_kind"):
        getattr(lib, q).restype = i32
        getattr(lib, q).argtypes = [vp]
    lib.hommx_comm_init_all.restype = C.c_int
    lib.hommx_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    lib.hommx_comm_destroy.restype = C.c_int
    lib.hommx_comm_destroy.argtypes = [vp]
    lib.hommx_comm_size.restype = C.c_int
    lib.hommx_comm_size.argtypes = [vp]
    lib.hommx_allgather_field.restype = C.c_int
    lib.hommx_allgather_field.argtypes = [vp, C.POINTER(vp), i64]
    lib.hommx_solve_batch_multi.restype = C.c_int
    lib.hommx_solve_batch_multi.argtypes = [vp, C.POINTER(vp), i64, vp, vp, vp, vp]
    lib.hommx_solve_batch_multi_device.restype = C.c_int
    lib.hommx_solve_batch_multi_device.argtypes = [vp, C.POINTER(vp), i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    i64p = C.POINTER(i64)
    lib.hommx_shard_range.restype = C.c_int
    lib.hommx_shard_range.argtypes = [i64, i32, i32, i64p, i64p, i64p]
    lib.hommx_unpack_field.restype = C.c_int
    lib.hommx_unpack_field.argtypes = [i64, i32, i32, vp, vp, vp]
    lib.hommx_calibrate_fp64.restype = C.c_int
    lib.hommx_calibrate_fp64.argtypes = [C.c_int, dp, dp]
    lib.hommx_calibrate_fp64_detail.restype = C.c_int
    lib.hommx_calibrate_fp64_detail.argtypes = [C.c_int, dp, dp, dp]
    lib.hommx_calibrate_fp64_mfma.restype = C.c_int
    lib.hommx_calibrate_fp64_mfma.argtypes = [C.c_int, dp]
    lib.hommx_last_error.restype = C.c_char_p
    lib.hommx_last_error.argtypes = []
    _lib = lib
    return lib


def last_error() -> str:
    return load().hommx_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise HommxLibraryError(f"{what} failed (code {rc}): {last_error()}")
