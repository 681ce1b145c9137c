#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HOMMX micro-cell hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: for every macro cell of the
configuration C2 of BASELINE.json (2D PoissonHMM, 64x64 macro mesh = 8192 triangles, 32x32 periodic
micro cells, inclusion coefficient) assemble the periodic micro problem, solve it, reduce to the
effective tensor A_H.  Inputs are resident in HBM before the timed region.  With N > 1 GPUs every rank
owns its own 8192-cell macro partition (weak scaling, the reference's MPI partition of hmm.py:307-310)
and the step ends with ONE RCCL all-gather of the effective-tensor field.

Prints ONE JSON line on rank 0 (contract: metric/value/unit/..., plus `roofline` and `cpu_baseline`).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

FP64_PEAK_DATASHEET = 78.6e12  # FLOP/s, AMD MI355X datasheet: FP64 vector == FP64 matrix (absent from the local guide)


def flop_model(n: int) -> float:
    """Dense block-cyclic model of what the fused kernel executes per micro-cell solve:
    per eliminated node row: symmetric sweep 2 b^3 + V' = W N 2 b^3 + S_last += V' W^T 2 b^3, b = n;
    n-1 such rows + the final sweep of the last block."""
    b = float(n)
    return (6.0 * (n - 1) + 2.0) * b**3


def algorithmic_bytes(n: int, stratified: bool) -> float:
    """coefficient stream + (M) + A_H."""
    return 8.0 * (2 * n * n + (4 if stratified else 0) + 4)


def cpu_baseline(coef: np.ndarray, n: int, vols_X: np.ndarray, budget_s: float = 15.0):
    """Reference-shaped CPU path (oracle, one core) on a bounded sample: cells coef[0], coef[1], ... until the budget."""
    from oracle import hommx_oracle as O

    t0 = time.perf_counter()
    done = 0
    AH = []
    while done < coef.shape[0]:
        S = O.local_stiffness_reference_shaped("poisson", 2, n, vols_X[done], coef[done], 2.0**-8)
        cp = O.build_cell_problem("poisson", 2, n, coef[done])
        AH.append(O.effective_tensor(cp))
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, np.stack(AH)


def cpu_worker(args):
    """Child process of the multi-core CPU baseline: `bench.py --cpu-worker START COUNT` runs the one-core loop on its own
    slice of the same workload (the reference partitions the macro cells over MPI ranks the same way, hmm.py:307-310)
    and prints {"done", "seconds"}.  Never touches the GPU."""
    from hommx_amd import workloads

    start, count = args.cpu_worker
    msh, coef_h, _ = workloads.c2_inclusion(args.macro, args.micro)
    X = msh.cell_vertices()
    rate, done, _ = cpu_baseline(coef_h[start : start + count], args.micro, X[start : start + count], args.cpu_budget)
    print(json.dumps({"done": done, "seconds": done / rate}))


def cpu_baseline_multicore(args, coef_h, n, X, cores: int):
    """`cores` one-core loops side by side (this process is one of them and keeps its tensors for the parity check).
    Rate = all cells done / the slowest worker's time."""
    import subprocess

    nc = coef_h.shape[0]
    per = nc // cores
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    kids = [
        subprocess.Popen(
            [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(w * per), str(per), "--macro", str(args.macro),
             "--micro", str(n), "--cpu-budget", str(args.cpu_budget)],
            stdout=subprocess.PIPE, text=True, env=env)
        for w in range(1, cores)
    ]
    rate0, done0, AH = cpu_baseline(coef_h[:per], n, X[:per], args.cpu_budget)
    done, slowest = done0, done0 / rate0
    for k in kids:
        out, _ = k.communicate(timeout=600)
        r = json.loads(out.strip().splitlines()[-1])
        done += r["done"]
        slowest = max(slowest, r["seconds"])
    return done / slowest, done, done0, AH


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--macro", type=int, default=64, help="macro cells per side (C2: 64)")
    ap.add_argument("--micro", type=int, default=32, help="micro cells per side (C2: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the CPU baseline (0: min(16, available))")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work per core")
    ap.add_argument("--cpu-worker", type=int, nargs=2, metavar=("START", "COUNT"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun the RCCL path is exercised even with one rank
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hommx_amd import MicroCellPlan, workloads
    from hommx_amd.dist import all_gather_field

    n = args.micro
    # rank r owns the macro partition [r, r+1] x [0, 1] of a (world x 1) strip of unit squares
    msh, coef_h, _ = workloads.c2_inclusion(args.macro, n, x_shift=float(rank))
    nc = coef_h.shape[0]
    plan = MicroCellPlan(2, n, "poisson", device=local_rank)
    coef = torch.from_numpy(coef_h).to(dev)
    out = torch.empty(nc, 2, 2, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        plan.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), stream.cuda_stream)
        if ev is not None:
            ev[1].record(stream)
        if use_dist:
            full = all_gather_field(out, world * nc)
            if ev is not None:
                ev[2].record(stream)  # the current stream waits for the collective: [1] -> [2] is the all-gather
            return full
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        field = step(evs[k])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))  # HIP events on the launch stream
    ag_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if use_dist else None
    n_bad = int((info != 0).sum().item())

    if rank == 0:
        value = world * nc * args.steps / dt
        flops = flop_model(n) * nc
        achieved = flops / (kern_ms * 1e-3)
        rec = {
            "metric": "micro-cell solves/sec",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "allgather_ms": ag_ms,  # RCCL all-gather of the A_H field per step; null without a process group
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2: 2D PoissonHMM, {args.macro}x{args.macro} macro P1 mesh ({nc} cells per GPU), "
                f"{n}x{n} periodic micro cells, inclusion coefficient",
                "cells_per_gpu": nc,
                "n_micro": n,
                "kernel": plan.kernel,
                "parallelism": f"macro-cell shards x{world}" + (", RCCL all-gather of A_H" if use_dist else ""),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved / 1e12,
                "peak": FP64_PEAK_DATASHEET / 1e12,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_DATASHEET,
                "traffic": None,
                "kernel": "k_poisson2d_fused<32>" if n > 16 else "k_poisson2d_fused<16>",
                "kernel_ms": kern_ms,
                "flops_per_solve": flop_model(n),
                "hbm_bytes_per_solve_algorithmic": algorithmic_bytes(n, False),
                "hbm_frac_algorithmic": algorithmic_bytes(n, False) * nc / (kern_ms * 1e-3) / 8.0e12,
            },
            "info_nonzero": n_bad,
        }
        # HBM traffic per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit in
        # one pass and must not be mixed with tracing), so the figure comes from the committed summary of exactly this
        # command (profiles/r01_i_pmc_summary.json: FETCH_SIZE x 2 as the hardware guide prescribes on gfx950, + WRITE_SIZE)
        try:
            pm = json.load(open(os.path.join(HERE, "profiles", "r01_i_pmc_summary.json")))
            if pm["cells_per_launch"] == nc and n == 32:
                rec["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                rec["roofline"]["traffic_source"] = "profiles/r01_i_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                rec["roofline"]["traffic_algorithmic"] = algorithmic_bytes(n, False) * nc
        except Exception:
            pass
        # fp64 MFMA calibration (the local guide lists no f64 matrix peak)
        try:
            import ctypes

            from hommx_amd import _lib

            f = ctypes.c_double()
            _lib.check(_lib.load().hommx_calibrate_fp64_mfma(local_rank, ctypes.byref(f)), "calibrate")
            rec["roofline"]["peak_measured_mfma_f64"] = f.value / 1e12
        except Exception as e:  # pragma: no cover
            rec["roofline"]["peak_measured_mfma_f64"] = None
            print(f"[bench] calibration failed: {e}", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            X = msh.cell_vertices()
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = args.cpu_cores if args.cpu_cores > 0 else min(16, avail)
            rate, ntotal, ndone, AH_cpu = cpu_baseline_multicore(args, coef_h, n, X, cores)
            AH_gpu = field[:ndone].cpu().numpy()
            err = float(np.max(np.linalg.norm(AH_gpu - AH_cpu, axis=(1, 2)) / np.linalg.norm(AH_cpu, axis=(1, 2))))
            rec["cpu_baseline"] = {
                "value": rate,
                "unit": "solves/s",
                "cores": cores,
                "kind": "port",
                "sample": f"{ntotal} macro cells of the same batch in {cores} equal slices, one process per core "
                f"({args.cpu_budget:g} s budget each; the reference partitions cells over MPI ranks the same way); oracle "
                "restatement of hmm.py:334-369 (3 corrector solves + 9 energies per cell, SciPy splu); host cores "
                f"available to this process: {avail}",
            }
            rec["effective_tensor_max_rel_err_vs_oracle"] = err
        print(json.dumps(rec))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
