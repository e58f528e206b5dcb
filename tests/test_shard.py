"""CPU tests of the multi-GPU path: contiguous sharding and the final gather over a world_size-2
gloo group give exactly the unsharded logs."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT


def test_shard_range_partition():
    from nmpc_amd.shard import shard_range
    for total in (1, 7, 8, 4096, 16384, 65537):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from nmpc_amd.shard import shard_range, gather_logs, reduce_stats, env_world
    rank, world, _ = env_world()
    dist.init_process_group("gloo")
    total = 11
    full = {"u_log": torch.arange(total * 5, dtype=torch.float64).reshape(total, 5),
            "lpv_counter": (torch.arange(total * 5) %% 4 + 1).to(torch.int32).reshape(total, 5)}
    lo, hi = shard_range(total, rank, world)
    out = gather_logs({k: v[lo:hi].clone() for k, v in full.items()}, total=total)
    for k in full:
        assert torch.equal(out[k], full[k]), k
    # gather to one rank, shard lengths known to the caller (no size exchange)
    sizes = [b - a for a, b in (shard_range(total, r, world) for r in range(world))]
    out = gather_logs({k: v[lo:hi].clone() for k, v in full.items()}, dst=1, sizes=sizes)
    for k in full:
        assert torch.equal(out[k], full[k] if rank == 1 else full[k][lo:hi]), k
    s = reduce_stats([float(hi - lo), 1.0], "sum")
    assert s == [float(total), float(world)], s
    m = reduce_stats([float(rank)], "max")
    assert m == [float(world - 1)]
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
