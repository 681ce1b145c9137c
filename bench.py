#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HOMMX micro-cell hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input.

--config C2 (default; the configuration BASELINE.json's metric is quoted on): for every macro cell of C2 (2D PoissonHMM,
  64x64 macro mesh = 8192 triangles, 32x32 periodic micro cells, inclusion coefficient) assemble the periodic micro problem,
  solve it, reduce to the effective tensor A_H.  Inputs are resident in HBM before the timed region.  With N > 1 GPUs every rank
  owns its own 8192-cell macro partition (WEAK scaling, the reference's MPI partition of hmm.py:307-310) and the step ends with
  ONE RCCL all-gather of the effective-tensor field.
--config C5 (BASELINE config 5): 3D LinearElasticityStratifiedHMM, 32x16x8 macro box = 24,576 tets, 16^3 micro cells,
  rotated-fibre theta.  The 24,576 cells are block-partitioned over the N ranks (STRONG scaling); every rank samples only its
  own shard (two phase values + M per cell; the fibre mask once), solves it and the C_H field is all-gathered over RCCL.

Prints ONE JSON line on rank 0 (contract: metric/value/unit/..., plus `roofline` and `cpu_baseline`).

`value` is the device-resident rate the bench contract asks for (inputs in HBM when the timed region starts; equal to
`value_device_resident`).  SURVEY 8(d) defines the unit of work WITH the transfers ("H2D of inputs and D2H of A_eff included"): that
figure is `value_survey_8d` -- the C-ABI call `hommx_solve_batch_two_phase` on HOST arrays, C2's own coefficient shape (a 2 KB phase
mask + two values per macro cell in, A_H + info out, one synchronisation) -- and `value_host_boundary` is the generic 134 MB element
stream through `hommx_solve_batch` (PCIe-bound).  The default C2 line also carries a nested `"c5"` record (BASELINE config 5: the 3D
stratified-elasticity path on the nested-dissection route, all 24,576 cells, `--c5-steps` timed passes with the workspace reserved
outside the timed region) with its own `roofline` and `cpu_baseline`; `--no-c5` leaves it out.
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
PMC_SUMMARY = os.path.join("profiles", "r04_c2_pmc_summary.json")
MF_PMC_SUMMARY = os.path.join("profiles", "r04_mf_pmc_summary.json")


def flop_model(n: int, b: int | None = None) -> float:
    """Dense block-cyclic model of what the kernels execute per micro-cell solve:
    per eliminated node row / plane: inverse 2 b^3 + V' = W N 2 b^3 + S_last += V' W^T 2 b^3; n-1 of them + the last block."""
    b = float(n if b is None else b)
    return (6.0 * (n - 1) + 2.0) * b**3


def flops_executed_fused2d(n: int) -> float:
    """FMAs x 2 the fused 2D kernel issues per solve on its padded block size NB (csrc/fused2d.hip, sweep_acc.h): per eliminated node row
    the inverse (NB = 32: two 16-sweeps + 16 MFMAs of the 2 x 2 block inverse; NB = 16: one 16-sweep), V'^T = N W^T (all tiles) and
    S_last += V' W^T (tiles on and below the diagonal); + the inverse of the last block.  The sparse coupling and load-row work
    (vector unit, about 8 % more) is not counted."""
    if n > 16:
        inv, v, sl = 2 * 16 * 256 + 16 * 1024, 32 * 1024, 24 * 1024
    else:
        inv, v, sl = 16 * 256, 4 * 1024, 4 * 1024
    return 2.0 * ((n - 1) * (inv + v + sl) + inv)


def flops_ref(dim: int, n: int, bs: int, nrhs: int) -> dict:
    """F_ref of SURVEY 8(d): sparse-Cholesky flops under a fill-reducing ordering (tools/fref.py); the 3D figure takes a
    minute of symbolic analysis, so it is read from the committed profiles/fref.json when it is there."""
    try:
        table = json.load(open(os.path.join(HERE, "profiles", "fref.json")))
        for e in table.values():
            if (e["dim"], e["n"], e["bs"], e["nrhs"]) == (dim, n, bs, nrhs):
                return e
    except Exception:
        pass
    sys.path.insert(0, os.path.join(HERE, "tools"))
    import fref

    return fref.fref(dim, n, bs, nrhs)


def algorithmic_bytes(n: int, stratified: bool) -> float:
    """coefficient stream + (M) + A_H."""
    return 8.0 * (2 * n * n + (4 if stratified else 0) + 4)


def usable_cores() -> int:
    """Host cores this process can actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box shows 256 cores
    but grants a share of them)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = min(avail, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                avail = min(avail, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return avail


def cpu_baseline(coef: np.ndarray, n: int, vols_X: np.ndarray, budget_s: float = 15.0):
    """Reference-shaped CPU path (oracle, one core) on a bounded sample: cells coef[0], coef[1], ... until the budget.
    ONLY the reference-shaped local stiffness (hmm.py:334-369: nb corrector solves + nb^2 energies) is inside the timed loop."""
    from oracle import hommx_oracle as O

    t0 = time.perf_counter()
    done = 0
    while done < coef.shape[0]:
        O.local_stiffness_reference_shaped("poisson", 2, n, vols_X[done], coef[done], 2.0**-8)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done


def cpu_baseline_c5(shape, n: int, cells: np.ndarray):
    """Reference-shaped CPU path for C5 (oracle, one core): hmm.py:334-369 for the given macro cells -- 12 corrector solves on the
    12,288-unknown periodic elasticity problem (SciPy splu) + 144 energies per cell, about half a minute each."""
    from hommx_amd import workloads
    from oracle import hommx_oracle as O

    msh, mask, values, M = workloads.c5_two_phase(shape, n, cells=cells)
    X = msh.cell_vertices()[cells]
    t0 = time.perf_counter()
    for k in range(len(cells)):
        coef = np.where(mask[:, None], values[k, 1][None, :], values[k, 0][None, :])  # [n_el, (lambda, mu)]
        O.local_stiffness_reference_shaped("elasticity", 3, n, X[k], coef, 2.0**-8, M[k])
    dt = time.perf_counter() - t0
    return len(cells) / dt, len(cells)


def cpu_baseline_c5_multicore(args, n_sample: int, n: int):
    """"C4/C5 CPU: time a fixed subsample (first 8 cells) and extrapolate linearly".  One cell per process, side by
    side; rate = cells / the slowest worker's time."""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--config", "C5", "--cpu-worker", str(w), "1", "--micro",
                              str(n), "--c5-shape"] + [str(v) for v in args.c5_shape], stdout=subprocess.PIPE, text=True, env=env)
            for w in range(n_sample)]
    done, slowest = 0, 0.0
    for k in kids:
        out, _ = k.communicate(timeout=900)
        r = json.loads(out.strip().splitlines()[-1])
        done += r["done"]
        slowest = max(slowest, r["seconds"])
    return done / slowest, done, slowest


def cpu_worker(args):
    """Child process of the multi-core CPU baseline: `bench.py --cpu-worker START COUNT` runs the one-core loop on its own
    slice of the same workload (the reference partitions the macro cells over MPI ranks the same way, hmm.py:307-310)
    and prints {"done", "seconds"}.  Never touches the GPU."""
    from hommx_amd import workloads

    start, count = args.cpu_worker
    if args.config == "C5":
        rate, done = cpu_baseline_c5(tuple(args.c5_shape), args.micro, np.arange(start, start + count))
        print(json.dumps({"done": done, "seconds": done / rate}))
        return
    msh, coef_h, _ = workloads.c2_inclusion(args.macro, args.micro)
    X = msh.cell_vertices()
    rate, done = cpu_baseline(coef_h[start : start + count], args.micro, X[start : start + count], args.cpu_budget)
    print(json.dumps({"done": done, "seconds": done / rate}))


def cpu_baseline_multicore(args, coef_h, n, X, cores: int):
    """`cores` one-core loops side by side (this process is one of them).  Rate = all cells done / the slowest worker's time."""
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
    rate0, done0 = cpu_baseline(coef_h[:per], n, X[:per], args.cpu_budget)
    done, slowest = done0, done0 / rate0
    for k in kids:
        out, _ = k.communicate(timeout=600)
        r = json.loads(out.strip().splitlines()[-1])
        done += r["done"]
        slowest = max(slowest, r["seconds"])
    return done / slowest, done, rate0


def launch_mode(gpus: int, env, visible_devices) -> str:
    """How `bench.py --gpus N` must run (the reference's only parallel strategy is N ranks, each owning a macro partition:
    hmm.py:307-310, docs/usage/usage.md:64-71).  Returns "inline" (this process is the one rank, or one rank of a launcher's
    N) or "spawn" (plain `python bench.py --gpus N`: this process starts the N ranks itself).  Raises SystemExit -- never a
    silent run on fewer GPUs -- when the launcher's WORLD_SIZE disagrees with --gpus or fewer than N devices are visible.
    `visible_devices` is a callable so that nothing touches the GPU before the children exist."""
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}: need at least one GPU")
    launched = "RANK" in env or "WORLD_SIZE" in env
    if launched:
        world = int(env.get("WORLD_SIZE", "1"))
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but the launcher set WORLD_SIZE={world}: refusing to report {world} rank(s) as {gpus} GPUs")
        return "inline"
    if gpus == 1:
        return "inline"
    have = int(visible_devices())
    if have < gpus:
        raise SystemExit(f"--gpus {gpus} but only {have} device(s) visible: refusing to run on fewer GPUs than asked for")
    return "spawn"


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(gpus: int, argv: list[str], timeout_s: float = 3000.0) -> int:
    """Start `gpus` ranks of this script as CHILD processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set), relay rank 0's JSON line, return non-zero if any rank
    failed.  The parent never initialises the GPU and never execs: the ranks are ordinary children."""
    import subprocess

    port = _free_port()
    kids = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                     stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    try:
        out0, _ = kids[0].communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:  # a rank that never arrives leaves the others in a collective: end them all, fail loudly
        for k in kids:
            k.kill()
        print(f"[bench] ranks did not finish within {timeout_s:.0f} s: killed", file=sys.stderr)
        return 1
    rcs = [kids[0].returncode]
    for k in kids[1:]:
        try:
            rcs.append(k.wait(timeout=300))
        except subprocess.TimeoutExpired:  # rank 0 is gone: a survivor would wait in a collective for ever
            k.kill()
            rcs.append(-9)
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs) or line is None:
        print(f"[bench] ranks exited with {rcs}" + ("" if line else "; rank 0 printed no JSON line"), file=sys.stderr)
        return 1
    rec = json.loads(line)
    if rec.get("n_gpus") != gpus:
        print(f"[bench] asked for {gpus} GPUs, line says n_gpus={rec.get('n_gpus')}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def dry_run(args):
    """Launcher rehearsal WITHOUT a GPU (hidden flag --dry-run-cpu, used by tests/test_bench_launch.py): gloo ranks, the plan
    replaced by a stub that writes the rank into its shard; the emitted line has the contract's keys with value = null and
    "data": "dry-run", so it can never pass for a measurement."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from hommx_amd.dist import all_gather_field, shard_range

    if args.config == "C5":
        ntot = 6 * args.c5_shape[0] * args.c5_shape[1] * args.c5_shape[2]
        b, e, per = shard_range(ntot, rank, world)
        t, scaling = 6, "strong"
    else:
        per = 2 * args.macro * args.macro
        b, e, ntot, t, scaling = rank * per, (rank + 1) * per, world * per, 2, "weak"
    out = torch.zeros(per, t, t, dtype=torch.float64)
    out[: e - b] = float(rank + 1)
    t0 = time.perf_counter()
    if use_dist and args.config == "C5":  # the product's unpadding (dist._unpad_index): ragged shards, ranks that own nothing
        from hommx_amd.dist import _unpad_index

        field = all_gather_field(out, world * per)[torch.from_numpy(_unpad_index(ntot, per, world))]
    else:
        field = all_gather_field(out, world * per) if use_dist else out
    ag_ms = (time.perf_counter() - t0) * 1e3 if use_dist else None
    owner = [int(v) for v in field[:, 0, 0].tolist()]
    owners = sorted({v for v in owner if v > 0})
    if rank == 0:
        rec = {"metric": "micro-cell solves/sec", "value": None, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": scaling,
               "vs_baseline": None, "dtype": "f64", "data": "dry-run", "dry_run": True, "allgather_ms": ag_ms,
               "config": {"workload": f"{args.config} launcher rehearsal on gloo, no kernel", "cells_total": ntot},
               "ranks_seen_in_gathered_field": owners}
        if args.config == "C5":
            rec["cells_per_rank"] = [shard_range(ntot, r, world)[1] - shard_range(ntot, r, world)[0] for r in range(world)]
            rec["owner_of_cell"] = owner
        print(json.dumps(rec), flush=True)
    if use_dist:
        dist.destroy_process_group()


def timed_steps(step, steps, warmup, use_dist, dist, dev, torch):
    """The contract's timing: W untimed steps, barrier + synchronize, K steps, synchronize + barrier, MAX over ranks."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
    t0 = time.perf_counter()
    last = None
    for k in range(steps):
        last = step(evs[k])
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
    return dt, kern_ms, ag_ms, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 1000 for C2 -- 1.2 s of GPU time, long enough for an external utilisation sampler; 1 for C5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 20 for C2, 1 for C5)")
    ap.add_argument("--config", choices=("C2", "C5"), default="C2")
    ap.add_argument("--macro", type=int, default=64, help="C2: macro cells per side (64)")
    ap.add_argument("--micro", type=int, default=None, help="micro cells per side (C2: 32, C5: 16)")
    ap.add_argument("--c5-shape", type=int, nargs=3, default=(32, 16, 8), help="C5: macro box (32 16 8 = 24,576 tets)")
    ap.add_argument("--no-c5", action="store_true", help="C2: leave the nested C5 record (3D elasticity path) out of the line")
    ap.add_argument("--c5-steps", type=int, default=2, help="C2: timed passes over all 24,576 C5 cells in the nested record (about 9 s each on one GPU)")
    ap.add_argument("--c5-warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host-boundary and two-phase figures")
    ap.add_argument("--cpu-cores", type=int, default=0, help="host cores for the CPU baseline (0: min(16, available))")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work per core")
    ap.add_argument("--cpu-worker", type=int, nargs=2, metavar=("START", "COUNT"), help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)  # launcher rehearsal on gloo (tests)
    ap.add_argument("--dry-run-devices", type=int, default=None, help=argparse.SUPPRESS)  # pretended device count (tests)
    args = ap.parse_args()
    if args.micro is None:
        args.micro = 32 if args.config == "C2" else 16
    if args.steps is None:
        args.steps = 1000 if args.config == "C2" else 1
    if args.warmup is None:
        args.warmup = 20 if args.config == "C2" else 1  # C5: the first call allocates the plan's workspace (~100 GB of fronts)
    if args.cpu_worker:
        return cpu_worker(args)

    def visible_devices():
        if args.dry_run_devices is not None:
            return args.dry_run_devices
        import torch  # device_count() does not initialise the GPU on this image

        return torch.cuda.device_count()

    if launch_mode(args.gpus, os.environ, visible_devices) == "spawn":
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run_cpu:
        return dry_run(args)

    # stdout carries ONE JSON line: native libraries that chat on fd 1 (RCCL prints a version banner at communicator creation)
    # are sent to stderr for the duration of the run
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(rec):
        sys.stdout.flush()
        os.dup2(_stdout_fd, 1)
        print(json.dumps(rec), flush=True)
        os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus  # launch_mode() has made sure
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
    from hommx_amd.dist import all_gather_field, shard_range

    n = args.micro
    stream = torch.cuda.current_stream()
    if args.config == "C5":
        rec = run_c5(args, torch, dist, use_dist, dev, rank, local_rank, world, MicroCellPlan, workloads, all_gather_field,
                     shard_range, stream)
        if rank == 0:
            emit(rec)
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- C2 ------------------------------------------------------------------------------------------------------------
    # rank r owns the macro partition [r, r+1] x [0, 1] of a (world x 1) strip of unit squares
    msh, coef_h, _ = workloads.c2_inclusion(args.macro, n, x_shift=float(rank))
    nc = coef_h.shape[0]
    plan = MicroCellPlan(2, n, "poisson", device=local_rank)
    coef = torch.from_numpy(coef_h).to(dev)
    out = torch.empty(nc, 2, 2, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)

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

    dt, kern_ms, ag_ms, field = timed_steps(step, args.steps, args.warmup, use_dist, dist, dev, torch)
    n_bad = int((info != 0).sum().item())

    # nested record of BASELINE config 5 (the 3D elasticity path): every rank takes part (strong scaling of the 24,576 cells)
    c5 = None
    if not args.no_c5:
        try:
            c5 = run_c5(args, torch, dist, use_dist, dev, rank, local_rank, world, MicroCellPlan, workloads, all_gather_field,
                        shard_range, stream, steps=args.c5_steps, warmup=args.c5_warmup, n=16)
        except Exception as exc:  # the C2 line must not be lost to a failure of the nested run; run_c5 made the ranks agree on it
            print(f"[bench] nested C5 record failed: {type(exc).__name__}: {exc}", file=sys.stderr)
            c5 = {"error": f"{type(exc).__name__}: {exc}"}
        stream = torch.cuda.current_stream()

    if rank == 0:
        value = world * nc * args.steps / dt
        flops = flop_model(n) * nc
        achieved = flops / (kern_ms * 1e-3)
        fr = flops_ref(2, n, 1, 2)
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
                "flop_model": "algorithmic: dense block-cyclic elimination without symmetry savings, (6 (n-1) + 2) n^3 (DESIGN.md section 2; "
                "the model of every round, so `frac` compares across rounds)",
                # what the kernel really issues: lower tiles of S_last, 2 x 2 block inverse with transposed tiles
                "flops_executed_per_solve": flops_executed_fused2d(n),
                "frac_executed": flops_executed_fused2d(n) * nc / (kern_ms * 1e-3) / FP64_PEAK_DATASHEET,
                # the same rate priced with the flops a sparse Cholesky under a fill-reducing ordering needs
                "flops_ref_per_solve": fr["F_ref"],
                "flops_ref_ordering": fr["ordering"],
                "frac_ref": fr["F_ref"] * nc / (kern_ms * 1e-3) / FP64_PEAK_DATASHEET,
                "hbm_bytes_per_solve_algorithmic": algorithmic_bytes(n, False),
                "hbm_frac_algorithmic": algorithmic_bytes(n, False) * nc / (kern_ms * 1e-3) / 8.0e12,
            },
            "info_nonzero": n_bad,
        }
        rec["value_device_resident"] = value
        rec["value_note"] = ("`value` = device-resident rate (bench contract: inputs in HBM when the timed region starts); SURVEY 8(d)'s unit of work "
                             "includes H2D of the inputs and D2H of A_eff: `value_survey_8d`")
        if c5 is not None:
            rec["c5"] = c5
        # HBM traffic per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit in
        # one pass and must not be mixed with tracing), so the figure comes from the committed summary of exactly this
        # command (FETCH_SIZE x 2 as the hardware guide prescribes on gfx950 for 16 B/lane reads, + WRITE_SIZE)
        try:
            pm = json.load(open(os.path.join(HERE, PMC_SUMMARY)))
            if pm["cells_per_launch"] == nc and n == 32:
                rec["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                rec["roofline"]["traffic_source"] = PMC_SUMMARY + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                rec["roofline"]["traffic_algorithmic"] = algorithmic_bytes(n, False) * nc
        except Exception:
            pass
        # fp64 calibration (the local guide lists no f64 peak): best of the MFMA and the FMA loop over 2 and 4 waves per SIMD --
        # both instructions share one fp64 pipe, so the larger is the measured peak a kernel mixing them is priced against
        try:
            import ctypes

            from hommx_amd import _lib

            fm, ff, fl = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            _lib.check(_lib.load().hommx_calibrate_fp64_detail(local_rank, ctypes.byref(fm), ctypes.byref(ff), ctypes.byref(fl)), "calibrate")
            peak_m = max(fm.value, ff.value, fl.value)
            rec["roofline"]["peak_measured_mfma_f64"] = fm.value / 1e12  # dependency-free pure-MFMA loop (clocks down under its own load)
            rec["roofline"]["peak_measured_mfma_lds_fed_f64"] = fl.value / 1e12  # the MFMA fed from LDS like a GEMM: what a real kernel can hold
            rec["roofline"]["peak_measured_fma_f64"] = ff.value / 1e12
            rec["roofline"]["peak_measured_f64"] = peak_m / 1e12
            rec["roofline"]["frac_of_measured_peak"] = achieved / peak_m
            rec["roofline"]["frac_executed_of_measured_peak"] = flops_executed_fused2d(n) * nc / (kern_ms * 1e-3) / peak_m
        except Exception as e:  # pragma: no cover
            rec["roofline"]["peak_measured_f64"] = None
            print(f"[bench] calibration failed: {e}", file=sys.stderr)

        if world == 1 and not args.no_host_boundary:
            # SURVEY 8(d)'s unit of work with the transfers in it: pageable NumPy arrays through hommx_solve_batch
            reps = 5
            plan.solve(coef_h)  # staging buffers allocated
            t0 = time.perf_counter()
            for _ in range(reps):
                A_host = plan.solve(coef_h)
            th = (time.perf_counter() - t0) / reps
            rec["value_host_boundary"] = nc / th
            rec["host_boundary_ms"] = th * 1e3
            rec["host_boundary_note"] = (f"hommx_solve_batch on pageable host arrays: H2D of {coef_h.nbytes / 1e6:.0f} MB in 2048-cell "
                                         "chunks overlapped with the kernel, D2H of A_H and info; never `value`")
            _, mask_h, values_h = workloads.c2_inclusion_two_phase(args.macro, n)
            # timed at the C ABI itself (what a reference-side binding would call), on arrays prepared once: hommx_solve_batch_two_phase packs
            # mask + values into the plan's pinned block, one H2D, the kernel, one D2H, one synchronisation
            from hommx_amd import _lib as _l

            lib_ = _l.load()
            mask_u8 = np.ascontiguousarray(np.asarray(mask_h).astype(np.uint8))
            vals_c = np.ascontiguousarray(values_h, dtype=np.float64)
            A_tp = np.empty((nc, 2, 2))
            info_tp = np.zeros(nc, dtype=np.int32)
            call_tp = lambda: _l.check(lib_.hommx_solve_batch_two_phase(plan._h, nc, mask_u8.ctypes.data, vals_c.ctypes.data, None, A_tp.ctypes.data,
                                                                         info_tp.ctypes.data), "hommx_solve_batch_two_phase")
            call_tp()
            reps_tp = 50
            t0 = time.perf_counter()
            for _ in range(reps_tp):
                call_tp()
            tp = (time.perf_counter() - t0) / reps_tp
            rec["value_two_phase"] = nc / tp
            rec["value_survey_8d"] = nc / tp
            rec["value_survey_8d_note"] = ("SURVEY 8(d) unit of work, transfers included: hommx_solve_batch_two_phase on HOST arrays (C2's own coefficient "
                                           "shape: 2 KB phase mask + two values per macro cell in, A_H + info out, one synchronisation), mean of "
                                           f"{reps_tp} calls")
            rec["two_phase_ms"] = tp * 1e3
            rec["two_phase_over_kernel"] = tp * 1e3 / kern_ms
            rec["two_phase_bitwise_equal_to_stream"] = bool(np.array_equal(A_tp, A_host))

        if not args.no_cpu_baseline and world == 1:
            from oracle import hommx_oracle as O

            X = msh.cell_vertices()
            avail = usable_cores()
            cores = args.cpu_cores if args.cpu_cores > 0 else min(16, avail)
            rate, ntotal, rate_core0 = cpu_baseline_multicore(args, coef_h, n, X, cores)
            # parity sample, outside every timed region: oracle tensors of 16 cells spread over the batch
            sample = np.linspace(0, nc - 1, 16).astype(int)
            AH_cpu = np.stack([O.effective_tensor(O.build_cell_problem("poisson", 2, n, coef_h[c])) for c in sample])
            AH_gpu = field[:nc].cpu().numpy()[sample]
            err = float(np.max(np.linalg.norm(AH_gpu - AH_cpu, axis=(1, 2)) / np.linalg.norm(AH_cpu, axis=(1, 2))))
            rec["cpu_baseline"] = {
                "value": rate,
                "unit": "solves/s",
                "cores": cores,
                "per_core": rate / cores,
                "kind": "port",
                "sample": f"{ntotal} macro cells of the same batch in {cores} equal slices, one process per core "
                f"({args.cpu_budget:g} s budget each; the reference partitions cells over MPI ranks the same way); the timed loop "
                "holds only the oracle restatement of hmm.py:334-369 (3 corrector solves + 9 energies per cell, SciPy splu); "
                f"host cores usable by this process (affinity mask capped by the cgroup quota): {avail}",
            }
            rec["effective_tensor_max_rel_err_vs_oracle"] = err
            # SURVEY 8(d) baseline (ii): the optimised CPU path -- oracle/hommx_oracle_c.c (dense block-cyclic elimination in plain
            # C, x86-64-v3, OpenMP over the macro cells) on ALL host cores this process may use, the whole batch, best of 3
            try:
                from oracle import c_oracle

                c_oracle.effective_tensor_batch_c(n, coef_h[:64], threads=avail)  # thread pool up
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    AH_c = c_oracle.effective_tensor_batch_c(n, coef_h, threads=avail)
                    best = min(best or 1e30, time.perf_counter() - t0)
                used = c_oracle.effective_tensor_batch_c.threads_used
                rec["cpu_baseline_optimized"] = {
                    "value": nc / best,
                    "unit": "solves/s",
                    "cores": used,
                    "per_core": nc / best / used,
                    "kind": "port",
                    "sample": f"the whole {nc}-cell batch, best of 3; oracle/hommx_oracle_c.c: closed-form periodic stencil, dense "
                    "block-cyclic elimination (the scheme the GPU kernel runs), gcc -O3 -march=x86-64-v3 -fopenmp",
                    "max_rel_err_vs_gpu": float(np.max(np.linalg.norm(AH_c - field[:nc].cpu().numpy(), axis=(1, 2))
                                                       / np.linalg.norm(AH_c, axis=(1, 2)))),
                }
            except Exception as e:  # pragma: no cover
                print(f"[bench] optimised CPU baseline failed: {e}", file=sys.stderr)
        emit(rec)
    if use_dist:
        dist.destroy_process_group()


def run_c5(args, torch, dist, use_dist, dev, rank, local_rank, world, MicroCellPlan, workloads, all_gather_field, shard_range,
           stream, steps=None, warmup=None, n=None):
    """BASELINE config 5: strong scaling of the 24,576 stratified 3D elasticity cells over the ranks.  Returns the record on rank 0
    (None elsewhere).  Under a process group the ranks agree that every one of them got through its set-up (plan, workspace, inputs)
    BEFORE the first collective of the timed loop: a rank that failed there makes all of them raise instead of leaving the others
    waiting in an all-gather."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    n = args.micro if n is None else n
    shape = tuple(args.c5_shape)
    ntot = 6 * shape[0] * shape[1] * shape[2]
    b, e, per = shard_range(ntot, rank, world)
    cells = np.arange(b, e)
    setup_error = None
    try:
        _, mask_h, values_h, M_h = workloads.c5_two_phase(shape, n, cells=cells)  # this rank's shard only
        plan = MicroCellPlan(3, n, "elasticity", device=local_rank)
        t_res = time.perf_counter()
        plan.reserve(e - b)  # the workspace (fronts of a chunk of cells) is part of the plan, not of a timed step
        torch.cuda.synchronize()
        reserve_s = time.perf_counter() - t_res
        mask = torch.from_numpy(mask_h.astype(np.uint8)).to(dev)
        values = torch.from_numpy(values_h).to(dev)
        M = torch.from_numpy(M_h).to(dev)
        out = torch.zeros(per, 6, 6, dtype=torch.float64, device=dev)  # padded shard
        info = torch.zeros(per, dtype=torch.int32, device=dev)
    except Exception as exc:  # noqa: BLE001 - agreed on below
        setup_error = exc
    if use_dist:
        flag = torch.tensor([0.0 if setup_error is None else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0 and setup_error is None:
            setup_error = RuntimeError("C5 set-up failed on another rank")
    if setup_error is not None:
        raise setup_error
    nloc = e - b

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        if nloc:
            plan.solve_two_phase_device(nloc, mask.data_ptr(), values.data_ptr(), M.data_ptr(), out.data_ptr(), info.data_ptr(),
                                        stream.cuda_stream)
        if ev is not None:
            ev[1].record(stream)
        if use_dist:
            full = all_gather_field(out, world * per)
            if ev is not None:
                ev[2].record(stream)
            return full
        return out

    dt, kern_ms, ag_ms, field = timed_steps(step, steps, warmup, use_dist, dist, dev, torch)
    n_bad = int((info != 0).sum().item())
    if use_dist:
        nb = torch.tensor([float(n_bad)], dtype=torch.float64, device=dev)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        n_bad = int(nb.item())
    if rank != 0:
        return None
    bdim = 3 * n * n
    F = flop_model(n, bdim)
    fr = flops_ref(3, n, 3, 6)
    F_exec = plan.flops_per_solve  # what the route executes (nested dissection: sum over the fronts of s^3 + 2 s^2 r + s r^2, padded sizes)
    achieved = F_exec * nloc / (kern_ms * 1e-3)
    C = field[:nloc].cpu().numpy()
    sym = float(np.abs(C - np.transpose(C, (0, 2, 1))).max() / np.abs(C).max())
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        avail = usable_cores()
        n_sample = max(1, min(8, avail, ntot))
        rate, done, slowest = cpu_baseline_c5_multicore(args, n_sample, n)
        cpu = {
            "value": rate,
            "unit": "solves/s",
            "cores": n_sample,
            "per_core": rate / n_sample,
            "kind": "port",
            "sample": f"the first {done} macro cells of the same workload, one process per cell side by side ({slowest:.0f} s for the slowest), "
            "extrapolated linearly to the batch (SURVEY 8(d)); the timed loop holds only the oracle restatement of hmm.py:334-369 "
            "(12 corrector solves on 12,288 unknowns with SciPy splu + 144 energies per cell); host cores usable by this process: "
            f"{avail}",
        }
    rec = {
        "metric": "micro-cell solves/sec",
        "value": ntot * steps / dt,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "reserve_s": reserve_s,  # hommx_plan_reserve before the first step: the route's workspace (hipMalloc), outside the timed region
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "allgather_ms": ag_ms,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C5: 3D LinearElasticityStratifiedHMM, {shape[0]}x{shape[1]}x{shape[2]} macro box ({ntot} tets), {n}^3 "
            "periodic micro cells (12,288 unknowns), rotated-fibre theta, two-phase fibre coefficient sampled on the device",
            "cells_total": ntot,
            "cells_this_rank": nloc,
            "n_micro": n,
            "kernel": plan.kernel,
            "parallelism": f"block partition of the macro cells over {world} rank(s)" + (", RCCL all-gather of C_H" if use_dist else ""),
        },
        "roofline": {
            "bound": "mfma",
            "achieved": achieved / 1e12,
            "peak": FP64_PEAK_DATASHEET / 1e12,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_PEAK_DATASHEET,
            "traffic": None,
            "kernel": plan.route_detail,  # from the plan itself (hommx_plan_route_detail): route, tree, tile sizes
            "kernel_ms": kern_ms,
            "flops_per_solve": F_exec,
            "flop_model": "executed dense flops of the route by its own model (hommx_plan_flops_per_solve): nested dissection, sum over the "
            "fronts of the staged elimination + s r^2 on the PADDED front sizes -- `frac` therefore includes padding work and is not "
            "comparable with rounds 1-2, which priced the plane elimination's (6 (n-1) + 2) b^3 (flops_plane_model_per_solve / "
            "frac_plane_model); frac_ref prices the same time with F_ref and compares across rounds and routes",
            "flops_plane_model_per_solve": F,
            "frac_plane_model": F * nloc / (kern_ms * 1e-3) / FP64_PEAK_DATASHEET,
            "flops_ref_per_solve": fr["F_ref"],
            "flops_ref_ordering": fr["ordering"],
            "frac_ref": fr["F_ref"] * nloc / (kern_ms * 1e-3) / FP64_PEAK_DATASHEET,
        },
        "info_nonzero": n_bad,
        "symmetry_defect": sym,
    }
    # HBM traffic of the route: PMC counters need their own rocprofv3 passes, so the figure comes from the committed summary of
    # tools/profile_mf.sh (same problem size; FETCH_SIZE x 2 as the hardware guide prescribes on gfx950, + WRITE_SIZE), per solve
    try:
        pm = json.load(open(os.path.join(HERE, MF_PMC_SUMMARY)))
        if plan.kernel == "multifrontal" and n == 16:
            rec["roofline"]["traffic"] = pm["hbm_bytes_per_cell_total"] * nloc
            rec["roofline"]["traffic_per_solve"] = pm["hbm_bytes_per_cell_total"]
            rec["roofline"]["traffic_source"] = MF_PMC_SUMMARY + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over all kernels of the route)"
            rec["roofline"]["traffic_algorithmic_per_solve"] = 8.0 * (6 * n**3 * 2 + 9 + 36)  # coefficient stream + M + C_H
            rec["roofline"]["mfma_busy_fraction_gemm"] = pm["families"]["gemm_gather"]["mfma_busy_fraction_of_simd_cycles"]
            rec["roofline"]["hbm_GBps"] = pm["hbm_bytes_per_cell_total"] * nloc / (kern_ms * 1e-3) / 1e9
    except Exception:
        pass
    if cpu is not None:
        rec["cpu_baseline"] = cpu
    return rec


if __name__ == "__main__":
    main()
