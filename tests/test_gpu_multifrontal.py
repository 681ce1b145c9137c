"""Nested-dissection route of the blocked family (csrc/multifrontal.hip; -m gpu): the same micro problems as the plane elimination
(forms /root/reference/src/hommx/hmm.py:644-667 / 759-789 / 887-922 / 1024-1067 on the periodic unit cell, cell_problem.py:38-300),
another elimination order.  The route is chosen when the plan is created (plane block b >= HOMMX_MF_MIN_B, default 65: every block the one-launch kernels do not take), hence the child
processes for the forced / disabled variants."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("stage", ["352", "64"])
def test_small_meshes_against_the_oracle_every_kind(stage):
    """Forced onto the multifrontal route (b > 64): odd and even n, one and three unknowns per node, all four coefficient layouts, with and
    without the stratification matrix; trees of 10 to 60 fronts, padded and unpadded front sizes.  HOMMX_MF_STAGE = 64: fronts of 64 and
    more eliminated unknowns go through the staged elimination (two to five stages on these meshes) that production sizes use."""
    _child(f"""
        import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        from oracle import hommx_oracle as O
        from test_gpu_small_wave import _inputs, _oracle_args
        def spd_inputs(p, kind, nc, seed):   # 3D matrix-valued kinds: SPD by construction (G G^T + shift), in the layout of include/hommx_hip.h
            rng = np.random.default_rng(seed)
            coef, M = _inputs(p, kind, 3, nc, seed)
            if kind == "poisson_matrix":
                G = rng.uniform(-1, 1, size=(nc, p.n_el, 3, 3))
                A = 0.3 * G @ np.swapaxes(G, -1, -2) + 0.5 * np.eye(3)
                coef = np.stack([A[..., i, j] for i, j in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))], axis=-1)
            if kind == "elasticity_voigt":
                G = rng.uniform(-1, 1, size=(nc, p.n_el, 6, 6))
                C = 0.2 * G @ np.swapaxes(G, -1, -2) + np.diag([2.0, 2.0, 2.0, 0.5, 0.5, 0.5])
                iu = np.triu_indices(6)
                coef = C[..., iu[0], iu[1]]
            return coef, M
        for kind, dim, n in (("elasticity", 3, 5), ("elasticity", 3, 6), ("elasticity", 3, 7), ("elasticity_voigt", 3, 5), ("poisson", 3, 9),
                             ("poisson", 3, 11), ("poisson", 3, 12), ("poisson_matrix", 3, 9), ("elasticity", 3, 8)):
            p = MicroCellPlan(dim, n, kind)
            assert p.kernel == "multifrontal", (kind, n, p.kernel)
            coef, M = spd_inputs(p, kind, 4, 7)
            A, info = p.solve(coef, M, return_info=True)
            assert not info.any(), (kind, n, info)
            okind, ocoef = _oracle_args(O, kind, dim, coef)
            ref = O.effective_tensor_batch(okind, dim, n, ocoef[:2], M[:2])
            assert np.abs(A[:2] - ref).max() <= 1e-10 * np.abs(ref).max(), (kind, n, np.abs(A[:2] - ref).max() / np.abs(ref).max())
            A0 = p.solve(coef[:1])
            ref0 = O.effective_tensor_batch(okind, dim, n, ocoef[:1], None)
            assert np.abs(A0 - ref0).max() <= 1e-10 * np.abs(ref0).max()
            assert np.abs(A - np.swapaxes(A, 1, 2)).max() <= 1e-11 * np.abs(A).max()
        print("ok")
    """, {"HOMMX_MF_MIN_B": "65", "HOMMX_MF_STAGE": stage})


def test_bad_cells_are_flagged_and_do_not_leak_and_chunks_do_not_change_bits():
    """A cell with a negative coefficient fails a pivot check somewhere in the tree: its info is the (1-based) group of the failing front,
    every other cell of the batch is untouched -- also when the batch is cut into workspace chunks (HOMMX_BLOCKED_MEM_GB)."""
    code = f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        rng = np.random.default_rng(11)
        p = MicroCellPlan(3, 8, "elasticity")
        assert p.kernel == "multifrontal"
        coef = rng.uniform(0.5, 3.0, size=(37, p.n_el, 2))
        M = np.eye(3)[None] + 0.2 * rng.standard_normal((37, 3, 3))
        good, info = p.solve(coef, M, return_info=True)
        assert not info.any()
        bad = coef.copy(); bad[5] = -1.0; bad[30, ::7] = np.nan
        A, info = p.solve(bad, M, return_info=True)
        assert info[5] > 0 and info[30] > 0 and (info != 0).sum() == 2, info
        keep = info == 0
        assert np.array_equal(A[keep], good[keep])
        np.save(sys.argv[1], good)
        print("ok")
    """
    outs = []
    for tag, gb in (("small", "0.02"), ("default", None)):  # 1.3 MB per cell at n = 8: 0.02 GB = 15 cells per chunk
        f = os.path.join("/tmp", f"hommx_mf_chunk_{tag}_{os.getpid()}.npy")
        env = {"HOMMX_MF_MIN_B": "65"}
        if gb:
            env["HOMMX_BLOCKED_MEM_GB"] = gb
        subprocess_code = code.replace("sys.argv[1]", repr(f))
        _child(subprocess_code, env)
        outs.append(np.load(f))
        os.remove(f)
    assert np.array_equal(outs[0], outs[1])


def test_pieces_on_one_to_four_streams_give_the_same_bits():
    """A chunk runs as two to four pieces side by side on as many streams (mf_solve; HOMMX_MF_STREAMS = 1: the caller's stream alone).
    520 cells of a 5^3 elasticity mesh: four pieces by default -- bitwise the tensors of one stream, with a bad cell in the third piece
    flagged and nothing else touched, and four sampled cells against the oracle."""
    code = f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        rng = np.random.default_rng(5)
        p = MicroCellPlan(3, 5, "elasticity")
        assert p.kernel == "multifrontal"
        coef = rng.uniform(0.5, 3.0, size=(520, p.n_el, 2))
        coef[300] = -1.0
        M = np.eye(3)[None] + 0.2 * rng.standard_normal((520, 3, 3))
        A, info = p.solve(coef, M, return_info=True)
        assert info[300] > 0 and (info != 0).sum() == 1, np.nonzero(info)
        np.savez(sys.argv[1], A=A, coef=coef[[0, 131, 262, 519]], M=M[[0, 131, 262, 519]])
        print("ok")
    """
    outs = []
    for streams in ("4", "1"):
        f = os.path.join("/tmp", f"hommx_mf_streams_{streams}_{os.getpid()}.npz")
        _child(code.replace("sys.argv[1]", repr(f)), {"HOMMX_MF_STREAMS": streams})
        outs.append(dict(np.load(f)))
        os.remove(f)
    keep = np.arange(520) != 300
    assert np.array_equal(outs[0]["A"][keep], outs[1]["A"][keep])
    sys.path.insert(0, ROOT)
    from oracle import hommx_oracle as O

    ref = O.effective_tensor_batch("elasticity", 3, 5, outs[0]["coef"], outs[0]["M"])
    got = outs[0]["A"][[0, 131, 262, 519]]
    assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()


def test_production_size_agrees_with_the_plane_elimination():
    """16^3 micro cells, three unknowns per node: the two eliminations of the blocked family on the SAME cells (C4 fibre contrast 1e5 and
    moderate random media) -- different orders of 12,288 unknowns, results equal to rounding."""
    code = f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from hommx_amd import MicroCellPlan, workloads as W
        p = MicroCellPlan(3, 16, "elasticity")
        assert p.kernel == sys.argv[2], p.kernel
        rng = np.random.default_rng(3)
        coef = rng.uniform(0.5, 3.0, size=(9, p.n_el, 2))
        M = np.eye(3)[None] + 0.2 * rng.standard_normal((9, 3, 3))
        A, info = p.solve(coef, M, return_info=True)
        assert not info.any()
        msh, mask, values, _ = W.c4_two_phase(cells=np.array([0, 2000, 4319]))
        B, info = p.solve_two_phase(mask, values, None, return_info=True)
        assert not info.any()
        np.savez(sys.argv[1], A=A, B=B)
        print("ok")
    """
    res = {}
    for route, env in (("multifrontal", {}), ("blocked", {"HOMMX_MF_MIN_B": "0"})):
        f = os.path.join("/tmp", f"hommx_mf_vs_blocked_{route}_{os.getpid()}.npz")
        _child(code.replace("sys.argv[1]", repr(f)).replace("sys.argv[2]", repr(route)), env)
        res[route] = dict(np.load(f))
        os.remove(f)
    for key, tol in (("A", 1e-11), ("B", 1e-7)):  # contrast 1e5 in the fibre cells: conditioning, not the method
        a, b = res["multifrontal"][key], res["blocked"][key]
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (key, np.abs(a - b).max() / np.abs(b).max())


def test_correctors_on_the_same_plan_do_not_disturb_the_tensors(rng):
    """Effective tensors and correctors of one plan come from two multifrontal plans (the corrector plan keeps every front for the back
    substitution, hommx_solve_batch_correctors) with workspaces of different chunk sizes -- interleaved calls must not see each other's
    buffers, and the corrector call returns the same tensors."""
    from hommx_amd import MicroCellPlan

    p = MicroCellPlan(3, 6, "elasticity")
    assert p.kernel == "multifrontal"
    coef = rng.uniform(0.5, 3.0, size=(40, p.n_el, 2))
    M = np.eye(3)[None] + 0.2 * rng.standard_normal((40, 3, 3))
    A1 = p.solve(coef, M)
    A2, chi = p.solve(coef[:3], M[:3], return_correctors=True)   # corrector plan, 3 cells: a much smaller workspace
    A3 = p.solve(coef, M)                                        # 40 cells on the tensor plan again
    assert np.array_equal(A1, A3)
    assert np.abs(A2 - A1[:3]).max() <= 1e-11 * np.abs(A1).max()
    assert chi.shape == (3, 6, 6**3 * 3) and np.isfinite(chi).all()


_CORR_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from hommx_amd import MicroCellPlan
kind, dim, n, nc, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rng = np.random.default_rng(11)
p = MicroCellPlan(dim, n, kind)
if kind == "poisson":
    coef = np.exp(rng.uniform(np.log(0.1), np.log(5.0), size=(nc, p.n_el)))
else:
    coef = np.stack([rng.uniform(0.5, 2.0, (nc, p.n_el)), np.exp(rng.uniform(np.log(0.1), np.log(10.0), (nc, p.n_el)))], axis=-1)
M = np.eye(dim)[None] + 0.3 * rng.standard_normal((nc, dim, dim))
A, chi, info = p.solve(coef, M, return_info=True, return_correctors=True)
np.savez(out, A=A, chi=chi, info=info, coef=coef, M=M, kernel=p.kernel)
"""


def _corr_child(kind, dim, n, nc, env):
    import os, subprocess, sys, tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "c.npz")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", _CORR_CHILD % {"root": root}, kind, str(dim), str(n), str(nc), f], check=True, env=e, timeout=900)
        return dict(np.load(f))


def _center(chi, bs):
    x = chi.reshape(chi.shape[0], -1, bs)
    return (x - x.mean(axis=1, keepdims=True)).reshape(chi.shape)


@pytest.mark.parametrize("kind,dim,n,env", [
    ("elasticity", 3, 5, {}),                                  # b = 75: the smallest 3D elasticity mesh of the route
    ("elasticity", 3, 6, {"HOMMX_MF_STAGE": "64"}),            # several stages inside the fronts: the back substitution walks them backwards
    ("poisson", 3, 10, {}),                                    # scalar 3D
    ("poisson", 2, 72, {"HOMMX_MF_MIN_B": "65"}),              # 2D: narrow fronts
    ("elasticity", 2, 36, {"HOMMX_MF_STREAMS": "1"}),          # one stream
])
def test_correctors_of_the_multifrontal_route_against_the_oracle(kind, dim, n, env):
    """hommx_solve_batch_correctors on a multifrontal plan: back substitution down the elimination tree (multifrontal.hip) against the
    oracle's correctors (hmm.py:397-432: K chi = b per load case, constants projected out) and, through the energy functional of the
    reference (hmm.py:652-667 / 905-922), against the effective tensors of the same call.  33 cells: two halves on two streams."""
    from oracle import hommx_oracle as O

    r = _corr_child(kind, dim, n, 33, env)
    assert str(r["kernel"]) == "multifrontal" and np.all(r["info"] == 0)
    bs = 1 if kind == "poisson" else dim
    for c in (0, 16, 17, 32):  # both halves
        cp = O.build_cell_problem(kind, dim, n, r["coef"][c], r["M"][c])
        chi = _center(O.solve_correctors(cp).T, bs)
        assert np.abs(r["chi"][c] - chi).max() < 1e-9 * np.abs(chi).max(), (c, np.abs(r["chi"][c] - chi).max() / np.abs(chi).max())
        AH = O.effective_tensor(cp, r["chi"][c].T, form="energy")
        assert np.abs(AH - r["A"][c]).max() < 1e-10 * np.abs(r["A"][c]).max()


def test_correctors_of_the_two_eliminations_agree_at_the_production_size():
    """16^3 elasticity cells (12,288 unknowns, BASELINE C4 / C5 size): correctors of the multifrontal route == correctors of the plane
    elimination (HOMMX_MF_CORR=0) on the same inputs."""
    a = _corr_child("elasticity", 3, 16, 3, {})
    b = _corr_child("elasticity", 3, 16, 3, {"HOMMX_MF_CORR": "0"})
    assert np.all(a["info"] == 0) and np.all(b["info"] == 0)
    assert np.abs(a["A"] - b["A"]).max() <= 1e-10 * np.abs(b["A"]).max()
    assert np.abs(a["chi"] - b["chi"]).max() <= 1e-8 * np.abs(b["chi"]).max(), np.abs(a["chi"] - b["chi"]).max() / np.abs(b["chi"]).max()


def test_stratified_elasticity_solver_class_on_the_multifrontal_route():
    """The reference's rotated-fibre example shape (LinearElasticityStratifiedHMM, examples/linear_elasticity/rotated_fibers.py:23-115,
    README.md:171-185) with 6^3 micro cells -- plane block b = 108, the nested-dissection route -- and the two-phase device sampler:
    final macro solution == the same class with the oracle standing in for the GPU plan (rel. L2 <= 1e-7 at fibre contrast 1e5)."""
    from hommx_amd import fem, hmm, mesh, workloads as W
    from test_hmm_host import with_oracle

    def Dt(x):
        x = np.asarray(x, float)
        if x.ndim == 1:
            return W.c5_theta_transpose(x[None, :3])[0]
        return np.moveaxis(W.c5_theta_transpose(x.T), 0, -1)

    def mk():
        msh = mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], (5, 2, 1))
        A = hmm.TwoPhase(lambda y: W.wrapped_disc(y[1], y[2]), lambda x: hmm.Lame(1.0 + 0 * x[0], 100.0 * (1.0 + x[0])),
                         lambda x: hmm.Lame(1.0 + 0 * x[0], 0.001 + 0 * x[0]))
        h = hmm.LinearElasticityStratifiedHMM(msh, A, lambda x: np.array([0.0, 0.0, -0.05 * 0.4**2]), mesh.create_unit_cube(6, 6, 6), 2.0**-5, Dt)
        V = h.function_space
        clamp = fem.locate_dofs_topological(V, 2, fem.locate_entities_boundary(msh, 2, lambda x: np.isclose(x[0], 0)))
        h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, V))
        return h

    h = mk()
    u = h.solve()
    assert h._plan.kernel == "multifrontal" and not h.cell_info.any()
    ref = with_oracle(mk()).solve()
    d = u.x.array - ref.x.array
    assert np.linalg.norm(d) <= 1e-7 * np.linalg.norm(ref.x.array), np.linalg.norm(d) / np.linalg.norm(ref.x.array)
    assert u.x.array.reshape(-1, 3)[:, 2].min() < 0  # the beam bends down


def test_random_sizes_kinds_contrasts_against_the_oracle():
    """Seeded sweep over the route's whole domain -- 2D and 3D, one to three unknowns per node, odd / even / prime n, contrasts up to 1e4,
    with and without M: every tree shape the symbolic analysis produces on these meshes against the oracle (observed <= 1e-14)."""
    _child(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        from oracle import hommx_oracle as O
        rng = np.random.default_rng(1)
        for it in range(14):
            dim = int(rng.choice([2, 3], p=[0.3, 0.7]))
            kind = str(rng.choice(["poisson", "elasticity"]))
            if dim == 3:
                n = int(rng.integers(5, 11)) if kind == "elasticity" else int(rng.integers(9, 14))
            else:
                n = int(rng.integers(66, 100)) if kind == "poisson" else int(rng.integers(34, 60))
            p = MicroCellPlan(dim, n, kind)
            assert p.kernel == "multifrontal", (dim, kind, n, p.kernel)
            contrast = 10 ** rng.uniform(0.5, 4)
            coef = 0.01 * np.exp(rng.uniform(0, np.log(contrast), size=(3, p.n_el) + ((2,) if kind == "elasticity" else ())))
            M = np.eye(dim)[None] + 0.3 * rng.standard_normal((3, dim, dim)) if rng.random() < 0.7 else None
            A, info = p.solve(coef, M, return_info=True)
            ref = O.effective_tensor(O.build_cell_problem(kind, dim, n, coef[0], None if M is None else M[0]))
            err = np.abs(A[0] - ref).max() / np.abs(ref).max()
            assert not info.any() and err < 1e-10, (dim, kind, n, contrast, err)
        print("ok")
    """, {"HOMMX_MF_MIN_B": "65"})


_FRONT_CASES = [("poisson", 2, 64), ("poisson", 2, 100), ("poisson", 2, 120), ("elasticity", 2, 28), ("elasticity", 2, 40), ("elasticity", 2, 62),
                ("poisson_matrix", 2, 72),
                ("poisson", 3, 9), ("poisson", 3, 12), ("elasticity", 3, 5), ("elasticity", 3, 8), ("elasticity_voigt", 2, 40)]


def test_register_resident_front_kernel_equals_the_launch_sequence_it_replaces(tmp_path):
    """csrc/mf_front_kernel.h (round 4): groups of small fronts are built, eliminated and reduced to their update matrix in ONE launch with
    the front in registers.  Same tree, same arithmetic up to the order of the sums: compared with the launch sequence it replaces
    (HOMMX_MF_FRONT=0: k_mf_build / k_mf_pad / recursive inverse / GEMMs) on every variant of the kernel -- one wave (T <= 4, T <= 6), four
    waves (T <= 8, T <= 12), eight waves (T <= 19; T = 20 / 21 with one / two tile rows in LDS: 2D Poisson 120^2 and 2D elasticity 62^2 have a
    T = 20 level, 3D elasticity 8^3 the T = 21 leaf of C4 / C5); one, two and three unknowns per node; leaf fronts and fronts with children; with
    smaller leaves (HOMMX_MF_LEAF=12) the 3D-elasticity tree of the C4 / C5 size class runs its two lowest levels on it -- and against the
    oracle.  A bad cell is flagged by the front kernel's pivot check and does not leak."""
    code = f"""
        import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np
        from hommx_amd import MicroCellPlan
        from test_gpu_small_wave import _inputs
        from test_gpu_multifrontal import _FRONT_CASES
        out = {{}}
        for kind, dim, n in _FRONT_CASES:
            p = MicroCellPlan(dim, n, kind)
            assert p.kernel == "multifrontal", (kind, dim, n, p.kernel)
            coef, M = _inputs(p, kind, dim, 6, 11)
            coef[4] = -np.abs(coef[4])          # not SPD: info must say so, the other cells must not notice
            A, info = p.solve(coef, M, return_info=True)
            assert info[4] > 0 and not np.delete(info, 4).any(), (kind, dim, n, info)
            out[f"{{kind}}_{{dim}}_{{n}}"] = np.delete(A, 4, axis=0)
        np.savez(sys.argv[1], **out)
        print("ok")
    """
    res = {}
    for tag, env in (("front", {}), ("sequence", {"HOMMX_MF_FRONT": "0"}), ("front_leaf12", {"HOMMX_MF_LEAF": "12"})):
        f = str(tmp_path / f"{tag}.npz")
        _child(code.replace("sys.argv[1]", repr(f)), env)
        res[tag] = dict(np.load(f))
    for key in res["front"]:
        a, b, c = res["front"][key], res["sequence"][key], res["front_leaf12"][key]
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (key, np.abs(a - b).max() / np.abs(b).max())
        assert np.abs(c - b).max() <= 1e-11 * np.abs(b).max(), (key, np.abs(c - b).max() / np.abs(b).max())
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hommx_amd import MicroCellPlan
    from oracle import hommx_oracle as O
    from test_gpu_small_wave import _inputs, _oracle_args

    for kind, dim, n in (("poisson", 2, 64), ("elasticity", 2, 28), ("elasticity", 3, 5)):
        p = MicroCellPlan(dim, n, kind)
        coef, M = _inputs(p, kind, dim, 6, 11)
        okind, ocoef = _oracle_args(O, kind, dim, coef)
        ref = O.effective_tensor_batch(okind, dim, n, ocoef[:2], M[:2])
        got = res["front"][f"{kind}_{dim}_{n}"][:2]
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), (kind, dim, n)
