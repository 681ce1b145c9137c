"""`bench.py --gpus N` must run N ranks however it is started, and never report fewer silently.

The reference's only parallel strategy is N ranks each owning a macro partition (/root/reference/src/hommx/hmm.py:307-310,
docs/usage/usage.md:64-71).  The launcher logic is rehearsed here on the CPU: gloo ranks, the plan replaced by a stub
(`--dry-run-cpu`; the emitted line carries value = null and "data": "dry-run").
"""

import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _run(argv, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)


def test_launch_mode_table():
    none = lambda: pytest.fail("device count must not be asked for")  # noqa: E731
    assert bench.launch_mode(1, {}, none) == "inline"
    assert bench.launch_mode(4, {"RANK": "2", "WORLD_SIZE": "4"}, none) == "inline"
    assert bench.launch_mode(2, {}, lambda: 2) == "spawn"
    assert bench.launch_mode(2, {}, lambda: 8) == "spawn"
    with pytest.raises(SystemExit, match="only 1 device"):
        bench.launch_mode(2, {}, lambda: 1)
    with pytest.raises(SystemExit, match="WORLD_SIZE=1"):
        bench.launch_mode(2, {"RANK": "0", "WORLD_SIZE": "1"}, none)
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        bench.launch_mode(1, {"RANK": "0", "WORLD_SIZE": "4"}, none)
    with pytest.raises(SystemExit):
        bench.launch_mode(0, {}, none)


@pytest.mark.parametrize("config", ["C2", "C5"])
def test_plain_invocation_starts_n_ranks(config):
    """`python bench.py --gpus 2` with no launcher environment: two ranks run, rank 0's line says n_gpus == 2 and the gathered
    field holds both ranks' shards."""
    extra = ["--macro", "4"] if config == "C2" else ["--c5-shape", "2", "1", "1"]
    r = _run(["--gpus", "2", "--config", config, "--dry-run-cpu", "--dry-run-devices", "2", "--steps", "1", "--warmup", "0"] + extra)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["value"] is None
    assert rec["ranks_seen_in_gathered_field"] == [1, 2]
    assert rec["scaling"] == ("weak" if config == "C2" else "strong")


def test_more_gpus_than_devices_fails_loudly():
    r = _run(["--gpus", "4", "--dry-run-cpu", "--dry-run-devices", "2"])
    assert r.returncode != 0
    assert "only 2 device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--dry-run-cpu"], {"RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_under_a_launcher_runs_inline():
    """torch.distributed.run's environment with one rank: no children, the line says n_gpus == 1."""
    r = _run(["--gpus", "1", "--dry-run-cpu", "--macro", "4"],
             {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(bench._free_port())})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ranks_seen_in_gathered_field"] == [1]


def test_eight_ranks_ragged_c5_shape():
    """eight gloo ranks on a ragged C5 shape (6 * 3 * 1 * 1 = 18 cells, 18 % 8 != 0: shards of 3 cells, the last two
    ranks own 0 cells) with the stub plan: the line says n_gpus == 8, every cell of the field was written exactly once by the rank
    that owns it, empty ranks took part in the collective, and the all-gather time is reported."""
    r = _run(["--gpus", "8", "--config", "C5", "--dry-run-cpu", "--dry-run-devices", "8", "--steps", "1", "--warmup", "0",
              "--c5-shape", "3", "1", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["dry_run"] is True and rec["value"] is None and rec["scaling"] == "strong"
    assert rec["config"]["cells_total"] == 18
    assert rec["cells_per_rank"] == [3, 3, 3, 3, 3, 3, 0, 0]
    assert rec["owner_of_cell"] == [1 + c // 3 for c in range(18)]   # every shard present once, in order, no padding leaked
    assert rec["ranks_seen_in_gathered_field"] == [1, 2, 3, 4, 5, 6]
    assert rec["allgather_ms"] is not None and rec["allgather_ms"] >= 0.0
