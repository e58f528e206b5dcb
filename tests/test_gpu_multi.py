"""Two ranks of the benchmark's multi-process path on ONE GPU (gloo rendezvous, both ranks on cuda:0):
exercises process-group set-up, the barriers around the timed region, the max-over-ranks reduction and the
final gather of the logs exactly as `torch.distributed.run ... bench.py --gpus N` runs them on a node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--batch", "512", "--latency-steps", "2", "--rehearse-on-one-gpu"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 12 and r["warmup"] == 3 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] - 2 * 512 * 12 / (r["ms_per_step"] * 12e-3)) < 1e-6 * r["value"]
    assert "cpu_baseline" not in r                # host baseline belongs to single-GPU runs
    assert 1.0 <= r["mean_lpv_passes_per_step"] <= 4.0
