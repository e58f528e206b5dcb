"""The benchmark's multi-process path on ONE GPU.

`python bench.py --gpus 2` starts its own two ranks (fresh child processes); with --rehearse-on-one-gpu both
use cuda:0 and rendezvous over gloo (two ranks cannot share a GPU under RCCL).  This exercises the spawn,
process-group set-up, the barriers around the timed region, the max-over-ranks reduction and the final
gather exactly as a node runs them.  The RCCL code path itself (backend "nccl": device-tensor gather and
all-reduce) runs in a one-rank group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=400):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_rank_bench_spawned_by_bench_itself():
    r = run_bench("--gpus", "2", "--steps", "12", "--warmup", "3", "--batch", "512", "--latency-steps", "2", "--repeats", "2",
                  "--rehearse-on-one-gpu")
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["steps"] == 12 and r["warmup"] == 3 and r["scaling"] == "weak"
    assert r["config"]["controllers_total"] == 1024
    assert r["value"] > 0 and abs(r["value"] - 2 * 512 * 12 / (r["ms_per_step"] * 12e-3)) < 1e-6 * r["value"]
    assert "cpu_baseline" not in r                # host baseline belongs to single-GPU runs
    assert 1.0 <= r["mean_lpv_passes_per_step"] <= 4.0 and len(r["ms_per_step_repeats"]) == 2


def test_strong_scaling_configs_shard_a_fixed_total():
    r = run_bench("--gpus", "2", "--config", "cfg4", "--steps", "6", "--warmup", "2", "--batch", "600", "--latency-steps", "0",
                  "--repeats", "1", "--rehearse-on-one-gpu")
    assert r["n_ranks_seen"] == 2 and r["scaling"] == "strong" and r["config"]["controllers_total"] == 600
    assert r["config"]["batch_per_gpu"] == 300 and r["mean_lpv_passes_per_step"] > 1.5
    r = run_bench("--gpus", "2", "--config", "cfg5", "--steps", "1", "--warmup", "1", "--batch", "512", "--repeats", "1",
                  "--rehearse-on-one-gpu")
    assert r["n_ranks_seen"] == 2 and r["gathered_rows"] == 512 and r["qp_status_nonzero"] == 0


def test_rccl_gather_and_allreduce_run_on_backend_nccl():
    """One rank, backend nccl: the final exchange goes through RCCL's gather / all-reduce on device tensors."""
    r = run_bench("--collectives-at-world-1", "--steps", "8", "--warmup", "2", "--batch", "256", "--latency-steps", "0",
                  "--repeats", "1", "--no-cpu-baseline")
    assert r["collective_backend"] == "nccl" and r["n_ranks_seen"] == 1 and r["n_gpus"] == 1
