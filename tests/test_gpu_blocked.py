"""Parity of the generic blocked HIP path (3D, elasticity, matrix-valued A, stratified) against the CPU oracle (-m gpu).

Tolerance 1e-9 relative per cell tensor (float64; observed 1e-15..1e-12); high-contrast fibre cases 1e-7.
"""

import glob
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9


def relerr(A, ref):
    return float(np.max(np.linalg.norm(A - ref, axis=(1, 2)) / np.linalg.norm(ref, axis=(1, 2))))


@pytest.fixture(scope="module")
def O():
    from oracle import hommx_oracle

    return hommx_oracle


def plan(dim, n, kind, flags=0):
    from hommx_amd import MicroCellPlan

    return MicroCellPlan(dim, n, kind, flags=flags)


def test_all_golden_vectors():
    files = sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(f).startswith("fullsize_"))
    assert len(files) == 10
    for f in files:
        g = np.load(f)
        M = g["M"] if g["M"].size else None
        p = plan(int(g["dim"]), int(g["n"]), str(g["kind"]), flags=1)  # flags=1: force the blocked family
        assert p.kernel in ("small_wave", "small_fused", "blocked")  # the family's route for this plane block size
        A, info = p.solve(g["coef"], M, return_info=True)
        assert np.all(info == 0)
        assert relerr(A, g["A_eff"]) < TOL, f


@pytest.mark.parametrize("dim,n", [(2, 7), (2, 12), (3, 3), (3, 5)])
def test_matrix_valued_poisson(dim, n, rng, O):
    """Matrix-valued A (the reference's forms hmm.py:644-667 take any A: `A_micro * grad`)."""
    n_el = (2 if dim == 2 else 6) * n**dim
    nc = 3
    Q = rng.standard_normal((nc, n_el, dim, dim))
    A = np.einsum("ceij,cekj->ceik", Q, Q) + 0.2 * np.eye(dim)
    pairs = [(0, 0), (1, 1), (0, 1)] if dim == 2 else [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
    coef = np.stack([A[..., i, j] for i, j in pairs], axis=-1)
    M = np.eye(dim)[None] + 0.3 * rng.standard_normal((nc, dim, dim))
    for MM in (None, M):
        got = plan(dim, n, "poisson_matrix").solve(coef, MM)
        assert relerr(got, O.effective_tensor_batch("poisson", dim, n, A, MM)) < TOL


@pytest.mark.parametrize("dim,n", [(2, 6), (3, 3)])
def test_general_hooke_tensor_voigt(dim, n, rng, O):
    """Full (anisotropic) Hooke tensor through the 21/6-component Voigt kind."""
    from hommx_amd.hmm import hooke_to_voigt

    n_el = (2 if dim == 2 else 6) * n**dim
    nc = 2
    t = dim * (dim + 1) // 2
    # random SPD elasticity tensors with the minor and major symmetries
    E = O.unit_strains(dim)
    L = rng.standard_normal((nc, n_el, t, t))
    Cv = np.einsum("cemk,cenk->cemn", L, L) + 0.5 * np.eye(t)
    # tensorial basis dual to E^m: C = sum_mn Cd[m,n] Ed^m (x) Ed^n with Ed = E scaled so that E^m:Ed^n = delta
    scale = np.array([1.0 / np.sum(E[m] * E[m]) for m in range(t)])
    Ed = E * scale[:, None, None]
    C = np.einsum("cemn,mij,nkl->ceijkl", Cv, Ed, Ed)
    assert np.allclose(hooke_to_voigt(C, dim), Cv)
    iu = np.triu_indices(t)
    coef = Cv[..., iu[0], iu[1]]
    M = np.eye(dim)[None] + 0.25 * rng.standard_normal((nc, dim, dim))
    for MM in (None, M):
        got = plan(dim, n, "elasticity_voigt").solve(coef, MM)
        assert relerr(got, O.effective_tensor_batch("elasticity", dim, n, C, MM)) < TOL


def test_constant_hooke_and_layered_kats():
    """test_integration_linear_elasticity.py:205-322 (C_H = C) and the layered closed forms of SURVEY 8(c)."""
    for dim, n in ((2, 10), (3, 3), (3, 8)):
        n_el = (2 if dim == 2 else 6) * n**dim
        CH = plan(dim, n, "elasticity").solve(np.tile([1.25, 1.0], (1, n_el, 1)))[0]
        t = CH.shape[0]
        ref = np.zeros((t, t))
        ref[:dim, :dim] = 1.25 + 2.0 * np.eye(dim)
        ref[dim:, dim:] = np.eye(t - dim)
        assert np.abs(CH - ref).max() < 1e-12
    from hommx_amd import workloads as W

    n = 8
    yb = W.element_barycentres(3, n)
    mu = np.where(np.cos(2 * np.pi * yb[:, 0]) < 0, 5.0, 0.5)
    CH = plan(3, n, "elasticity").solve(np.stack([np.ones_like(mu), mu], axis=-1)[None])[0]
    assert abs(CH[0, 0] - 1.0 / np.mean([1 / 11.0, 1 / 2.0])) < 1e-11
    assert abs(CH[3, 3] - 1.0 / np.mean([1 / 5.0, 1 / 0.5])) < 1e-11


def test_poisson_3d_reference_case(rng, O):
    """test_integration_poisson.py:243-294: A = 1.1 + x0 + sin(2 pi y0) on 6^3 micro cells."""
    n = 6
    x0 = np.linspace(0.05, 0.95, 5)
    coef = np.stack([O.sample_coefficient(lambda x, y: 1.1 + x[0] + np.sin(2 * np.pi * y[0]), np.array([v, 0, 0]), 3, n, 3)
                     for v in x0])
    got = plan(3, n, "poisson").solve(coef)
    assert relerr(got, O.effective_tensor_batch("poisson", 3, n, coef)) < TOL
    # laminate in y0: transverse directions see the arithmetic mean 1.1 + x0 exactly
    assert np.abs(got[:, 1, 1] - (1.1 + x0)).max() < 1e-12


def test_c4_c5_reduced_vs_oracle(O):
    """C4 / C5 of BASELINE.json on a reduced macro mesh and 8^3 micro cells (fibre contrast 1e5): oracle parity."""
    from hommx_amd import workloads as W

    msh, coef, _ = W.c4_fibre_beam(shape=(2, 1, 1), n=8)
    got, info = plan(3, 8, "elasticity").solve(coef, return_info=True)
    assert np.all(info == 0)
    assert relerr(got, O.effective_tensor_batch("elasticity", 3, 8, coef)) < 1e-7
    msh, coef, M = W.c5_rotated_fibres(shape=(2, 1, 1), n=8)
    got, info = plan(3, 8, "elasticity").solve(coef, M, return_info=True)
    assert np.all(info == 0)
    assert relerr(got, O.effective_tensor_batch("elasticity", 3, 8, coef, M)) < 1e-7


@pytest.mark.parametrize("name", ["c4_n16", "c5_n16_strat"])
def test_full_size_golden_cells_default_plan(name):
    """Production size (16^3 micro cells, 12288 unknowns, Bp = 768: the 128x128 GEMM tiles and the strip-form sparse products)
    through the DEFAULT plan -- no flags, default HOMMX_GEMM128_MIN -- against oracle tensors of three cells of C4 / C5
    (lowest / middle / highest fibre contrast; most / half / least rotated M).  Forms hmm.py:887-922 / 1024-1067, coefficient
    rotated_fibers.py:23-76.  Fixtures: tests/golden/fullsize_*.npz, made by tests/golden/make_golden.py (oracle, ~25 s per cell)."""
    sys.path.insert(0, GOLDEN)
    from make_golden import expand_fibre_fixture

    g = np.load(os.path.join(GOLDEN, f"fullsize_{name}.npz"))
    coef = expand_fibre_fixture(g)
    assert coef.shape == (3, 24576, 2)
    M = g["M"] if g["M"].size else None
    p = plan(3, 16, "elasticity")
    assert p.kernel == "multifrontal"  # b = 768: nested dissection (csrc/multifrontal.hip); the plane elimination is checked below
    C, info = p.solve(coef, M, return_info=True)
    assert np.all(info == 0)
    assert relerr(C, g["A_eff"]) < 1e-7, relerr(C, g["A_eff"])
    # the same cells through the two-phase entry point (mask + two (lambda, mu) pairs per cell): bit-identical
    mask = np.unpackbits(g["mask_bits"])[: 24576]
    C2 = p.solve_two_phase(mask, g["values"], M)
    assert np.array_equal(C, C2)


def test_c5_full_size_chunked_properties():
    """C5 at production size on a subset that spans several workspace chunks (HOMMX_BLOCKED_MEM_GB is read once per process, so
    the small budget is set in a child process): chunking must not change a single bit, tensors are symmetric positive
    definite, and the fixture cells come out as in the one-chunk run."""
    import subprocess, textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r})
        import numpy as np
        from hommx_amd import MicroCellPlan, workloads as W
        msh, coef, M = W.c5_rotated_fibres(shape=(8, 4, 2))      # 384 tets of the C5 family, same theta, same fibre
        sel = np.arange(0, 384, 3)[:100]
        p = MicroCellPlan(3, 16, "elasticity")
        C, info = p.solve(coef[sel], M[sel], return_info=True)
        assert not info.any()
        np.save(sys.argv[1], C)
        print("ok")
    """)
    outs = []
    for tag, gb in (("small", "1.5"), ("default", None)):
        env = dict(os.environ)
        if gb:
            # the default plan of this size runs the nested-dissection route: 206 MB of fronts, inverse scratch and stencil per cell
            # => 7 cells per chunk at 1.5 GB: 15 chunks for 100 cells, every chunk in two pieces on two streams
            env["HOMMX_BLOCKED_MEM_GB"] = gb
        f = os.path.join("/tmp", f"hommx_c5_chunk_{tag}_{os.getpid()}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
        outs.append(np.load(f))
        os.remove(f)
    a, b = outs
    assert np.array_equal(a, b)
    assert np.abs(a - np.transpose(a, (0, 2, 1))).max() < 1e-9 * np.abs(a).max()
    assert np.all(np.linalg.eigvalsh(0.5 * (a + np.transpose(a, (0, 2, 1)))) > 0)


def test_c4_full_size_properties():
    """C4 at full size: 20x6x6 box = 4320 tets, 16^3 micro cells (12288 unknowns per cell), size-independent properties."""
    from hommx_amd import workloads as W

    msh, coef, _ = W.c4_fibre_beam()
    assert coef.shape == (4320, 24576, 2)
    p = plan(3, 16, "elasticity")
    C, info = p.solve(coef, return_info=True)
    assert np.all(info == 0)
    assert np.abs(C - np.transpose(C, (0, 2, 1))).max() < 1e-9 * np.abs(C).max()
    ev = np.linalg.eigvalsh(0.5 * (C + np.transpose(C, (0, 2, 1))))
    assert np.all(ev > 0)
    # Voigt / Reuss bounds on the axial stiffness along the fibres (y0): Reuss <= C_H[00,00] <= Voigt
    lam, mu = coef[..., 0], coef[..., 1]
    voigt = np.mean(lam + 2 * mu, axis=1)
    reuss = 1.0 / np.mean(1.0 / (lam + 2 * mu), axis=1)
    assert np.all(C[:, 0, 0] <= voigt * (1 + 1e-10)) and np.all(C[:, 0, 0] >= reuss * (1 - 1e-10))
    # the fibre is uniform along y0, so the axial modulus is close to the Voigt average and grows with mu_in(x0)
    c = msh.cell_midpoints()
    order = np.argsort(c[:, 0])
    assert np.all(np.diff(C[order, 0, 0][:: 6 * 6 * 6]) >= -1e-9)
    # cells with the same x0 have the same coefficient => identical tensors
    same = np.isclose(c[:, 0], c[0, 0])
    assert np.abs(C[same] - C[0]).max() < 1e-9 * np.abs(C[0]).max()
    # linearity
    assert relerr(p.solve(2.0 * coef[:8]), 2.0 * C[:8]) < 1e-12


def test_blocked_equals_fused_on_2d_poisson(rng):
    n, nc = 24, 16
    coef = np.exp(rng.uniform(np.log(0.01), np.log(10.0), size=(nc, 2 * n * n)))
    M = np.eye(2)[None] + 0.3 * rng.standard_normal((nc, 2, 2))
    a = plan(2, n, "poisson").solve(coef, M)
    b = plan(2, n, "poisson", flags=1).solve(coef, M)
    assert relerr(a, b) < 1e-11


def test_info_reports_bad_cells():
    n = 4
    coef = np.ones((3, 6 * n**3, 2))
    coef[1, :, 1] = -1.0  # negative shear modulus: not SPD
    A, info = plan(3, n, "elasticity").solve(coef, return_info=True)
    assert info[0] == 0 and info[2] == 0 and info[1] > 0


def test_2d_poisson_beyond_the_fused_family(rng, O):
    """n_micro = 40 > 32: the plan falls back to the blocked family (b = 40, Bp = 64)."""
    n, nc = 40, 3
    coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(nc, 2 * n * n)))
    M = np.eye(2)[None] + 0.3 * rng.standard_normal((nc, 2, 2))
    p = plan(2, n, "poisson")
    assert p.kernel == "small_wave"
    assert relerr(p.solve(coef, M), O.effective_tensor_batch("poisson", 2, n, coef, M)) < TOL


def test_bitwise_reproducible(rng):
    """No atomics, fixed reduction order: two runs of either family give identical bits."""
    n = 32
    coef = rng.uniform(0.1, 3.0, size=(64, 2 * n * n))
    p = plan(2, n, "poisson")
    assert np.array_equal(p.solve(coef), p.solve(coef))
    coef3 = np.stack([rng.uniform(0.5, 2.0, (4, 6 * 4**3)), rng.uniform(0.1, 10.0, (4, 6 * 4**3))], axis=-1)
    q = plan(3, 4, "elasticity")
    assert np.array_equal(q.solve(coef3), q.solve(coef3))


_G128_CASES = (("elasticity", 3, 5), ("poisson", 2, 12), ("elasticity", 2, 9))


def _g128_inputs(p, dim, seed):
    rng = np.random.default_rng(seed)
    shape = (11, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
    return rng.uniform(0.3, 3.0, size=shape), np.eye(dim)[None] + 0.2 * rng.standard_normal((11, dim, dim))


def test_large_tile_gemm_on_partial_tiles(tmp_path):
    """Route every GEMM of small problems (Bp = 96 / 160: partial 128-tiles, lower-only tiles, all transposes) through the
    128x128 kernel -- which production sizes reach only for Bp >= 256 -- and compare with the oracle and the default route.
    The knob is read once per process, hence the child process."""
    import subprocess, sys, textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        from oracle import hommx_oracle as O
        from test_gpu_blocked import _G128_CASES, _g128_inputs
        for kind, dim, n in _G128_CASES:
            p = MicroCellPlan(dim, n, kind, flags=1)
            coef, M = _g128_inputs(p, dim, 5)
            A, info = p.solve(coef, M, return_info=True)
            assert not info.any()
            ref = O.effective_tensor_batch(kind, dim, n, coef[:3], M[:3])
            err = np.abs(A[:3] - ref).max() / np.abs(ref).max()
            assert err < 1e-11, (kind, dim, n, err)
            np.save({str(tmp_path)!r} + f"/{{kind}}_{{dim}}_{{n}}.npy", A)
        print("ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HOMMX_GEMM128_MIN="32"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    for kind, dim, n in _G128_CASES:
        p = plan(dim, n, kind, flags=1)
        coef, M = _g128_inputs(p, dim, 5)
        B = np.load(tmp_path / f"{kind}_{dim}_{n}.npy")
        assert np.abs(p.solve(coef, M) - B).max() <= 1e-12 * np.abs(B).max()


def test_small_fused_route_equals_hbm_route(tmp_path):
    """Plane blocks b <= 64 run in ONE LDS-resident launch (csrc/small_fused.h: BP = 32 / 48 / 64); HOMMX_NO_SMALL_FUSED (read when
    the plan is created, hence the child process) sends the same inputs through the HBM-resident kernels of the blocked family.
    Sizes of the reference's own tests: 2D elasticity 10 x 10 (test_integration_linear_elasticity.py:62-171), 3D Poisson 6^3
    (test_integration_poisson.py:243-294); plus b = 64 (3D Poisson 8^3), b = 48 (3D elasticity 4^3) and matrix-valued 2D Poisson."""
    import subprocess, textwrap

    cases = [("elasticity", 2, 10), ("poisson", 3, 6), ("poisson", 3, 8), ("elasticity", 3, 4), ("poisson_matrix", 2, 16),
             ("elasticity_voigt", 2, 7)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        for kind, dim, n in {cases!r}:
            p = MicroCellPlan(dim, n, kind, flags=1)
            rng = np.random.default_rng(11)
            shape = (9, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
            coef = rng.uniform(0.4, 2.5, size=shape)
            if kind == "poisson_matrix":
                coef[..., -1] = 0.2 * rng.uniform(-1, 1, size=shape[:2])
            if kind == "elasticity_voigt":  # SPD 3 x 3: diagonally dominant upper triangle (00, 01, 02, 11, 12, 22)
                coef[..., [1, 2, 4]] *= 0.1
            M = np.eye(dim)[None] + 0.2 * rng.standard_normal((9, dim, dim))
            A, info = p.solve(coef, M, return_info=True)
            assert not info.any(), (kind, dim, n)
            np.save(sys.argv[1] + f"/{{kind}}_{{dim}}_{{n}}.npy", A)
        print("ok")
    """)
    outs = {}
    for tag, env_extra in (("small", {}), ("hbm", {"HOMMX_NO_SMALL_FUSED": "1"})):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", code, str(d)], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
        outs[tag] = {c: np.load(d / f"{c[0]}_{c[1]}_{c[2]}.npy") for c in cases}
    for c in cases:
        assert relerr(outs["small"][c], outs["hbm"][c]) < 1e-10, c
