"""All five BASELINE.json configurations on one MI355X: micro-cell solves/s + parity against the closed forms / the oracle.

    python tools/bench_configs.py [--full-c5]      (writes one JSON document to stdout)

C1-C3 run on the fused 2D kernel, C4/C5 on the blocked path.  C5 (24,576 macro cells x 393 KB of (lambda, mu) samples) is
generated and solved in slabs of macro cells so that the host never holds more than ~1 GB of coefficients.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan, workloads as W

dev = torch.device("cuda:0")


def timed(plan, coef, M, reps=3):
    nc = coef.shape[0]
    dc = torch.from_numpy(np.ascontiguousarray(coef)).to(dev)
    dM = torch.from_numpy(np.ascontiguousarray(M)).to(dev) if M is not None else None
    out = torch.empty(nc, plan.t, plan.t, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: plan.solve_device(nc, dc.data_ptr(), dM.data_ptr() if dM is not None else None, out.data_ptr(), info.data_ptr(), st)
    plan.reserve(nc)  # blocked family: workspace ahead of the timed calls (hommx_plan_reserve)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, out.cpu().numpy(), int((info != 0).sum())


def relerr(A, ref):
    return float(np.max(np.linalg.norm(A - ref, axis=(1, 2)) / np.linalg.norm(ref, axis=(1, 2))))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--full-c5", action="store_true"); args = ap.parse_args()
    from oracle import hommx_oracle as O
    res = {}
    # C1
    msh, coef, _ = W.c1_laminate(); p = MicroCellPlan(2, 16, "poisson")
    dt, A, bad = timed(p, coef, None, 20)
    res["C1"] = dict(cells=coef.shape[0], n_micro=16, kernel=p.kernel, ms=dt * 1e3, solves_per_s=coef.shape[0] / dt, bad=bad,
                     max_rel_err_closed_form=relerr(A, W.c1_exact(msh)))
    # C2
    msh, coef, _ = W.c2_inclusion(); p = MicroCellPlan(2, 32, "poisson")
    dt, A, bad = timed(p, coef, None, 10)
    idx = np.linspace(0, coef.shape[0] - 1, 16).astype(int)
    res["C2"] = dict(cells=coef.shape[0], n_micro=32, kernel=p.kernel, ms=dt * 1e3, solves_per_s=coef.shape[0] / dt, bad=bad,
                     max_rel_err_oracle_16_cells=relerr(A[idx], O.effective_tensor_batch("poisson", 2, 32, coef[idx])))
    # C3
    msh, coef, M = W.c3_wavy_laminate()
    dt, A, bad = timed(p, coef, M, 5)
    res["C3"] = dict(cells=coef.shape[0], n_micro=32, kernel=p.kernel, ms=dt * 1e3, solves_per_s=coef.shape[0] / dt, bad=bad,
                     max_rel_err_closed_form=relerr(A, W.stratified_laminate_exact(M)))
    # C4
    msh, coef, _ = W.c4_fibre_beam(); p = MicroCellPlan(3, 16, "elasticity")
    dt, A, bad = timed(p, coef, None, 1)
    res["C4"] = dict(cells=coef.shape[0], n_micro=16, kernel=p.kernel, ms=dt * 1e3, solves_per_s=coef.shape[0] / dt, bad=bad,
                     flops_model_TF=(6 * 15 + 2) * 768.0**3 * coef.shape[0] / dt / 1e12, symmetric=float(np.abs(A - A.transpose(0, 2, 1)).max()))
    del coef
    # C5 in slabs
    shape = (32, 16, 8) if args.full_c5 else (8, 4, 2)
    from hommx_amd import mesh as _mesh
    msh = _mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
    ncell = msh.num_cells
    slab = 2048
    tot = 0.0; bad = 0; sym = 0.0
    _, coef_all, M_all = None, None, None
    for a in range(0, ncell, slab):
        b = min(ncell, a + slab)
        sub = _mesh.Mesh(msh.geometry, msh.topology, msh.cells[a:b], msh.shape)
        c = sub.cell_midpoints()
        coef = W.fibre_lame(c, 16, np.full(c.shape[0], 100.0))
        gam = 0.5 * np.pi * c[:, 1] / 0.4; dg = 0.5 * np.pi / 0.4
        Dth = np.zeros((c.shape[0], 3, 3)); Dth[:, 0, 0] = 1; Dth[:, 1, 1] = 1
        Dth[:, 2, 0] = -np.sin(gam); Dth[:, 2, 1] = dg * (-np.sin(gam) * c[:, 2] - np.cos(gam) * c[:, 0]); Dth[:, 2, 2] = np.cos(gam)
        M = np.transpose(Dth, (0, 2, 1)).copy()
        dt, A, bd = timed(p, coef, M, 1)
        tot += dt; bad += bd; sym = max(sym, float(np.abs(A - A.transpose(0, 2, 1)).max()))
    res["C5"] = dict(cells=ncell, macro_shape=list(shape), n_micro=16, kernel=p.kernel, ms=tot * 1e3, solves_per_s=ncell / tot, bad=bad,
                     flops_model_TF=(6 * 15 + 2) * 768.0**3 * ncell / tot / 1e12, symmetric=sym)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
