"""C-ABI library: loads, exports every symbol include/hommx_hip.h declares, fails loudly without a GPU (CPU only)."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from hommx_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from hommx_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "hommx_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hommx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "hommx_hip.h")).read()
    assert "torch" not in hdr.lower().replace("no torch types", "") and "at::" not in hdr


def test_plan_create_argument_checks(lib):
    from hommx_amd import _lib

    h = ctypes.c_void_p()
    for desc in (_lib.PlanDesc(4, 8, 0, 0, 0), _lib.PlanDesc(2, 8, 9, 0, 0), _lib.PlanDesc(2, 2, 0, 0, 0)):
        rc = lib.hommx_plan_create(ctypes.byref(h), ctypes.byref(desc))
        assert rc == -1 and not h.value
        assert lib.hommx_last_error()


def test_product_path_has_no_cpu_fallback(lib):
    """Without a GPU the plan must fail (ENODEV), not compute on the CPU."""
    import torch

    from hommx_amd import MicroCellPlan, _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.hommx_device_count() == 0
    with pytest.raises(_lib.HommxLibraryError):
        MicroCellPlan(2, 16, "poisson")


def test_package_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hommx_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle restatement", "").replace("the oracle", "").replace("CPU oracle", ""), (dirpath, f)


def test_shard_range():
    from hommx_amd.dist import shard_range

    for n, w in ((10, 4), (8192, 8), (7, 8), (0, 2)):
        cover = []
        for r in range(w):
            b, e, per = shard_range(n, r, w)
            assert e - b <= per
            cover += list(range(b, e))
        assert cover == list(range(n))


def test_workloads_shapes():
    from hommx_amd import workloads as W

    msh, coef, M = W.c1_laminate()
    assert coef.shape == (128, 512) and M is None
    msh, coef, M = W.c3_wavy_laminate(nx=4, n=8)
    assert coef.shape == (32, 128) and M.shape == (32, 2, 2)
    msh, coef, M = W.c5_rotated_fibres(shape=(2, 1, 1), n=4)
    assert coef.shape == (12, 384, 2) and M.shape == (12, 3, 3)
    assert np.isclose(msh.cell_volumes().sum(), 1.0 * 0.4 * 0.1)


@pytest.mark.parametrize("n_cells,ndev", [(10, 4), (8192, 8), (7, 8), (0, 2), (24576, 3), (5, 1)])
def test_c_abi_shard_range_and_unpack(lib, n_cells, ndev):
    """The partition / unpack index arithmetic of hommx_solve_batch_multi as plain host functions (no GPU): P > 1, ragged shards,
    more devices than cells.  Same bounds as hommx_amd.dist.shard_range (the reference's MPI ownership ranges, hmm.py:307-310)."""
    from hommx_amd.dist import shard_range

    i64 = ctypes.c_int64
    cover, per0 = [], None
    for i in range(ndev):
        b, e, per = i64(), i64(), i64()
        assert lib.hommx_shard_range(n_cells, ndev, i, ctypes.byref(b), ctypes.byref(e), ctypes.byref(per)) == 0
        assert (b.value, e.value, per.value) == shard_range(n_cells, i, ndev)
        per0 = per.value
        cover += list(range(b.value, e.value))
    assert cover == list(range(n_cells))
    assert lib.hommx_shard_range(5, 2, 2, None, None, None) == -1 and lib.hommx_shard_range(5, 0, 0, None, None, None) == -1
    if n_cells == 0:
        return
    tt = 4
    # gathered layout: device i's padded shard at rows [i * per, (i + 1) * per); cell k carries (k, k + 0.25, ...) and info = k % 3
    packed = np.full((ndev, per0, tt + 1), -5.0)
    for i in range(ndev):
        b, e, _ = shard_range(n_cells, i, ndev)
        for k in range(b, e):
            packed[i, k - b, :tt] = k + 0.25 * np.arange(tt)
            packed[i, k - b, tt] = k % 3
    A = np.empty((n_cells, tt))
    info = np.empty(n_cells, np.int32)
    assert lib.hommx_unpack_field(n_cells, ndev, tt, packed.ctypes.data, A.ctypes.data, info.ctypes.data) == 0
    assert np.array_equal(A, np.arange(n_cells)[:, None] + 0.25 * np.arange(tt)[None, :])
    assert np.array_equal(info, np.arange(n_cells) % 3)
