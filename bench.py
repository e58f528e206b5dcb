#!/usr/bin/env python3
"""Benchmark of the batched closed-loop LPV-NMPC step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5]

Default (cfg2) is the headline: closed-loop controller-steps/s, 4096 controllers per GPU, horizon
N = 10, unbalanced-disc encoder model, random-level references (BASELINE.json configs[1]).
  cfg3  long horizon Nc = 50, u in [-3, 3], y in [-1, 1], 4096 controllers per GPU (weak scaling)
  cfg4  noisy measurement (sigma 0.014 = SNR 30), LPV passes to convergence (tol 1e-5, cap 10),
        16384 controllers in total, sharded over the N GPUs (strong scaling)
  cfg5  synthetic wide model (nx 8, hidden 256, 20 lambda-nodes, Nc 20): open-loop LPV iteration
        (embedding, condensing, QP), 65536 controllers in total over the N GPUs (strong scaling)

With --gpus N > 1 and no torchrun environment the script starts its own N ranks (fresh child
processes, one per GPU, before this process touches a GPU) and forwards rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of the ranks.
Ranks own disjoint controllers; there is no collective inside the loop; trajectories are gathered to
rank 0 and a few statistics all-reduced once at the end (RCCL, backend "nccl").

A "step" is one closed-loop MPC step of the rank's whole batch: up to max LPV passes {embedding,
condensing, QP, prediction}, plant RK4 step, measurement, observer, warm-start shift.  References,
targets and state are resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STAGES = 20
# GEMM flops of one scheduling point (one horizon step of one controller) per quadrature node (H = 64):
# f-net 2H(nx+nu)+2H^2+2nx H^2+2nx H(nx+nu) = 25728, h-net 2H nx+2H^2+2ny H^2+2ny H nx = 16896.
FLOPS_PER_POINT_NODE = 25728 + 16896
NODES_AS_WRITTEN = 41      # unique Simpson nodes of 20 stages (the algorithmic count)
NODES_EVALUATED = 40       # the node at lambda = 0 is a constant of the net, evaluated once on the host
# The embedding kernel forms every fp32 product W x from three fp16 MFMA products
# (Wh xh + Wh xl + Wl xh, fp32 accumulate), so its matrix-pipe ceiling in algorithmic (fp32-equivalent)
# flops is one third of the dense fp16 MFMA peak (16 x the fp32 MFMA peak of 157.3 TFLOP/s).
FP32_MFMA_PEAK_TFLOPS = 157.3
FP16_MFMA_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS
SPLIT_PRODUCTS = 3
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 (vector = matrix rate; datasheet figure, the microarchitecture notes list none)
EVENT_EVERY = 2            # the instrumented region brackets the launches of every 2nd step with HIP event pairs
INSTRUMENTED_STEPS = 64    # steps of the instrumented region (run after the timed regions; never part of `value`)
REPEATS = 3                # the K-step timed region is run this many times, event-free; `value` is the first
PROFILES = os.path.join(ROOT, "profiles")

CONFIGS = {
    "cfg2": dict(Nc=10, ubox=(-4.0, 4.0), ybox=(-1.2, 1.2), sigma=0.0, max_it=4, tol=1e-1, batch=4096, scaling="weak"),
    "cfg3": dict(Nc=50, ubox=(-3.0, 3.0), ybox=(-1.0, 1.0), sigma=0.0, max_it=4, tol=1e-1, batch=4096, scaling="weak"),
    "cfg4": dict(Nc=10, ubox=(-4.0, 4.0), ybox=(-1.2, 1.2), sigma=0.014, max_it=10, tol=1e-5, batch=16384, scaling="strong"),
    "cfg5": dict(Nc=20, batch=65536, scaling="strong"),
}


def random_level_reference(n, nt_range, level_range, rng):
    """Reference.py:9-20 with a NumPy Generator (piecewise-constant random levels)."""
    out = np.zeros(0)
    remaining = n
    while True:
        steps = int(rng.integers(nt_range[0], nt_range[1] + 1))
        remaining -= steps
        level = int(rng.integers(int(level_range[0] * 10), int(level_range[1] * 10) + 1)) / 10
        out = np.concatenate([out, np.full(steps, level)])
        if remaining <= 0:
            return out[:n]


def controller_noise(gid, nsim, sigma):
    """Measurement noise of global controller `gid`: a counter-based generator keyed by the
    controller id, so a controller sees the same noise whichever rank owns it."""
    return sigma * np.random.Generator(np.random.Philox(key=1234 + int(gid))).standard_normal(nsim)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the fp64 NumPy oracle (a restatement of the reference's CPU path), timed on this
# host in a child process that never touches the GPU.
def _cpu_one(job):
    """One controller of the baseline: returns (per-step bucket seconds [n, 4])."""
    import torch
    torch.set_num_threads(1)
    from oracle.model import Model
    from oracle import lpv as olpv
    from oracle.nmpc import output_nmpc_linear, target_selector
    ref, n_steps, functorch = job
    om = Model.load()
    nc = CONFIGS["cfg2"]["Nc"]
    nodes, weights = olpv.merge_nodes(*olpv.quad_nodes("simpson", STAGES))
    yrn = np.clip((ref - om.y0) / om.ystd, (-1.2 - om.y0) / om.ystd, (1.2 - om.y0) / om.ystd)
    # target selection is pre-loop set-up in the reference too: keep it out of the timed part
    Xs, Us, _, _ = target_selector(om, yrn, n_steps + nc, (-4 - om.u0) / om.ustd, (4 - om.u0) / om.ustd,
                                   np.array([(-1.2 - om.y0) / om.ystd]), np.array([(1.2 - om.y0) / om.ystd]),
                                   nodes, weights, as_executed=True)
    timers = np.zeros((n_steps, 4))
    output_nmpc_linear(om, -4, 4, [-1.2], [1.2], np.zeros(2), np.diag([1000.0, 10.0]), 1.0, 0.1, STAGES, ref, Nc=nc,
                       Nsim=n_steps, max_iterations=4, targets=(Xs, Us), timers=timers,
                       embed_fn=olpv.embed_functorch if functorch else None)
    return timers


def cpu_baseline_main(args):
    """Child process: prints one JSON object with the baseline figures."""
    import multiprocessing as mp
    nthreads = int(os.environ.get("NMPC_BASELINE_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    nthreads = max(1, min(nthreads, os.cpu_count() or 1))
    nsl, steps1, stepsn = args.cpu_controllers, args.cpu_steps, args.cpu_steps_mt

    def stats(tm, wall, n_ctrl):
        per = tm.sum(-1).reshape(-1)
        return {"controller_steps_per_s": n_ctrl * tm.shape[-2] / wall, "step_ms_mean": float(per.mean() * 1e3),
                "step_ms_p50": float(np.median(per) * 1e3), "step_ms_max": float(per.max() * 1e3),
                "bucket_ms_mean": dict(zip(("getAB", "solve", "overhead", "sim"), (tm.reshape(-1, 4).mean(0) * 1e3).tolist()))}
    out = {"cores": nthreads, "host_cpus": os.cpu_count()}
    # cfg-1: the reference's own signal, one controller, Nsim = 190
    ref1 = np.clip(np.load(os.path.join(ROOT, "tests", "golden", "ref_randomLevelTime10_15Range-1_1Nsim200.npy"))[1], -1.2, 1.2)
    _cpu_one((ref1, 5, False))                                   # imports, first-touch
    t = time.perf_counter(); tm = _cpu_one((ref1, 190, False)); w = time.perf_counter() - t
    out["cfg1_1thread"] = stats(tm, tm.sum(), 1)
    out["cfg1_1thread"]["wall_incl_target_selection_s"] = w
    _cpu_one((ref1, 3, True))
    tm = _cpu_one((ref1, 190, True))
    out["cfg1_1thread_vmap_jacrev"] = stats(tm, tm.sum(), 1)
    # a slice of cfg-2: the first controllers of the benchmark's batch
    refs = [random_level_reference(stepsn + 10, [10, 15], [-1, 1], np.random.default_rng(1234 + b)) for b in range(nsl)]
    t = time.perf_counter()
    tms = np.stack([_cpu_one((refs[b], steps1, False)) for b in range(nsl)])
    out["cfg2_slice_1thread"] = stats(tms, tms.sum(), nsl)
    out["cfg2_slice_1thread"]["sample"] = "controllers 0..%d of the batch, first %d closed-loop steps" % (nsl - 1, steps1)
    if nthreads > 1:
        with mp.get_context("fork").Pool(nthreads) as pool:
            pool.map(_cpu_one, [(refs[b], 2, False) for b in range(nthreads)])        # start the workers
            t = time.perf_counter()
            tms = np.stack(pool.map(_cpu_one, [(refs[b], stepsn, False) for b in range(nsl)], chunksize=1))
            wall = time.perf_counter() - t
        out["cfg2_slice_nthreads"] = stats(tms, wall, nsl)
        out["cfg2_slice_nthreads"]["sample"] = ("controllers 0..%d, first %d steps, %d worker processes (one controller each at "
                                                "a time; wall time includes each worker's target selection)" % (nsl - 1, stepsn, nthreads))
    print("CPU_BASELINE " + json.dumps(out), flush=True)


def run_cpu_baseline(args):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-controllers", str(args.cpu_controllers),
           "--cpu-steps", str(args.cpu_steps), "--cpu-steps-mt", str(args.cpu_steps_mt)]
    # worker count = this job's CPU share (OMP_NUM_THREADS where the launcher sets it); every worker itself
    # runs single-threaded BLAS so that "1 thread" and "n threads" mean what they say
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    env = dict(os.environ, NMPC_BASELINE_THREADS=str(share), OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    for line in p.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    raise SystemExit("cpu baseline child failed:\n" + p.stdout[-2000:] + p.stderr[-2000:])


# ------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script (before this process has touched a GPU) and
    wait for them; rank 0 prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is None:
                    continue
                pending.remove(p)
                if c != 0 and rc == 0:
                    rc = c
                    for q in pending:              # a failed rank leaves the others stuck in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc if rc >= 0 else 1


def isolated_embed_tflops(nmpc_amd, enc, lam, w, npts, dev, reps=20, batches=5):
    """One full embedding pass (npts scheduling points) alone on the current stream: `batches` event-timed runs
    of `reps` back-to-back launches each, after a warm-up of the same length; returns (median, all) in TFLOP/s."""
    import torch
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(npts, 2, generator=g) * 2 - 1).to(dev)
    u = (torch.rand(npts, 1, generator=g) * 2 - 1).to(dev)
    xc = (torch.rand(npts, 2, generator=g) * 2 - 1).to(dev)
    A = torch.empty(npts, 2, 2, device=dev); B = torch.empty(npts, 2, 1, device=dev); C = torch.empty(npts, 1, 2, device=dev)
    for _ in range(reps):
        nmpc_amd.embed(enc, x, u, xc, lam, w, out=(A, B, C))
    res = []
    for _ in range(batches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            nmpc_amd.embed(enc, x, u, xc, lam, w, out=(A, B, C))
        b.record()
        torch.cuda.synchronize(dev)
        res.append(NODES_EVALUATED * FLOPS_PER_POINT_NODE * npts / (a.elapsed_time(b) / reps * 1e-3) / 1e12)
    return float(np.median(res)), res


def pmc_traffic(name):
    """HBM traffic per launch of a kernel from the committed PMC summary profiles/<name> (written by
    tools/pmc_summary.py --json from separate rocprofv3 --pmc passes of this command): FETCH_SIZE and WRITE_SIZE
    are in KB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes, so it is doubled."""
    path = os.path.join(PROFILES, name)
    try:
        d = json.load(open(path))
        fetch, write = d["FETCH_SIZE"]["mean"] * 1024.0, d["WRITE_SIZE"]["mean"] * 1024.0
    except (OSError, KeyError, ValueError):
        return None, None
    return 2.0 * fetch + write, {"source": "profiles/" + name, "fetch_size_kb_raw": d["FETCH_SIZE"]["mean"],
                                 "write_size_kb": d["WRITE_SIZE"]["mean"], "launches": d["FETCH_SIZE"]["n"],
                                 "correction": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half "
                                               "the bytes of wide coalesced reads); per launch, averaged over the launches "
                                               "of the profiled run"}


def wide_solve_flops(Nc, nx=2):
    """fp64 flops of one long-horizon condense + QP solve with an empty working set (n = Nc + 1 variables,
    2 Nc + 1 row pairs): Gamma recurrences, G = Gamma' Omega Gamma over the lower block triangle, Cholesky of G,
    the transform L^-1 [pair normals], prediction; active-set iterations add 4 n (n + pairs) each."""
    n, pairs = Nc + 1, 2 * Nc + 1
    gam = Nc * (Nc + 1) / 2 * 2 * nx * nx
    hess = sum((a + 1) * (Nc - a) for a in range(Nc)) * (2 * nx * nx + 4 * nx)
    chol = n ** 3 / 3.0
    trans = n * n * pairs
    pred = Nc * (Nc + 1) / 2 * 2 * nx
    return gam + hess + chol + trans + pred, 4.0 * n * (n + pairs)


def solve_bytes_per_controller_pass(Nc, nx=2):
    """Algorithmic HBM bytes of one condense+QP launch per active controller: A, B, C (fp32), targets,
    plan and state in; plan, prediction, next scheduling points, status out."""
    rd = Nc * nx * nx * 4 + 2 * Nc * nx * 4 + Nc * nx * 8 + Nc * 8 + Nc * 8 + nx * 8 + 8
    wr = Nc * 8 + Nc * 4 + Nc * nx * 8 + 2 * Nc * nx * 4 + 24
    return rd + wr


def closed_loop_bench(args, cfgname, rank, world, dev, shard):
    import torch
    import torch.distributed as dist
    import nmpc_amd
    cfg = CONFIGS[cfgname]
    NC = cfg["Nc"]
    K, W = args.steps, args.warmup
    LAT = max(0, args.latency_steps)
    R = max(1, args.repeats)
    KI = 0 if args.no_kernel_events else max(0, min(args.instrumented_steps, K))
    nsim = W + R * K + KI + LAT
    if cfg["scaling"] == "weak":
        B = args.batch or cfg["batch"]
        lo = rank * B
        total = world * B
        sizes = [B] * world
    else:
        total = args.batch or cfg["batch"]
        lo, hi = shard.shard_range(total, rank, world)
        B = hi - lo
        sizes = [b - a for a, b in (shard.shard_range(total, r, world) for r in range(world))]
    refs = np.stack([random_level_reference(nsim + NC, [10, 15], [-1, 1], np.random.default_rng(1234 + lo + b))
                     for b in range(B)])
    enc = nmpc_amd.load_system(device=dev)
    nm = enc.norm
    lam, w = nmpc_amd.quad_nodes("simpson", STAGES)
    (ulo, uhi), (ylo, yhi) = cfg["ubox"], cfg["ybox"]
    ymin_n, ymax_n = (ylo - nm.y0) / nm.ystd, (yhi - nm.y0) / nm.ystd
    umin_n, umax_n = (ulo - nm.u0) / nm.ustd, (uhi - nm.u0) / nm.ustd
    yrn = torch.clamp((torch.as_tensor(refs, device=dev) - nm.y0) / nm.ystd, ymin_n, ymax_n)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    Xs, Us, _ = nmpc_amd.getXsUs_Cs(yrn, 2, 1, 1, nsim + NC, umin_n, umax_n, [ymin_n], [ymax_n],
                                    nmpc_amd.GetABC(enc, lam, w))
    torch.cuda.synchronize(dev)
    t_targets = time.perf_counter() - t0
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[cfg["sigma"]])
    stamps = None
    if args.solve_stamps:          # diagnostics: per-phase clock stamps of every controller's last condense+QP solve
        stamps = torch.zeros(B, 16, dtype=torch.int64, device=dev)
        os.environ["NMPC_MPC_STAMPS"] = hex(stamps.data_ptr())
    ctrl = nmpc_amd.Controller(enc, plant, B, NC, nsim, cfg["max_it"], np.diag([1000.0, 10.0]), 1.0, umin_n, umax_n,
                               ymin_n, ymax_n, lam, w, tol=cfg["tol"], rho=1000.0)
    noise = None
    if cfg["sigma"] > 0:
        noise = torch.as_tensor(np.stack([controller_noise(lo + b, nsim, cfg["sigma"]) for b in range(B)], axis=1), device=dev)
    ctrl.set_targets(Xs.transpose(1, 2).contiguous(), Us[:, 0, :].contiguous(), noise)
    ctrl.reset((np.zeros(2) - nm.y0) / nm.ystd, (0 - nm.u0) / nm.ustd)

    ctrl.run(0, W)                                   # untimed warm-up steps
    torch.cuda.synchronize(dev)
    every = 0 if not KI else max(1, args.event_every)
    region_ms = []
    for r in range(R):                               # region 0 is the reported one; no events, no stamps in any of them
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctrl.run(W + r * K, K)                       # EXACTLY K timed steps
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        region_ms.append((time.perf_counter() - t0) * 1e3)
    elapsed = region_ms[0] * 1e-3
    # instrumented region (separate from the timed ones): HIP event pairs on the launch streams around the launches
    # of every `every`-th step, and the embedding workgroups stamp the shader clock they ran at
    k_inst0 = W + R * K
    prof = {"embed": (0, 0.0), "solve": (0, 0.0), "plant": (0, 0.0)}
    inst_ms, clock_ghz = None, None
    if KI:
        ctrl.profile(every)
        stamp_ok = NC < 24 and cfg["max_it"] >= 1
        if stamp_ok:
            ctrl.embed_clock(True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctrl.run(k_inst0, KI)
        torch.cuda.synchronize(dev)
        inst_ms = (time.perf_counter() - t0) * 1e3
        prof = ctrl.profile_read()
        ctrl.profile(False)
        if stamp_ok:
            clock_ghz = ctrl.embed_clock(False)
    # per-step latency: one step of the whole batch at a time, host-synchronised (not part of `value`)
    lat = []
    for k in range(W + R * K + KI, W + R * K + KI + LAT):
        t1 = time.perf_counter()
        ctrl.run(k, 1)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)
    p50_latency = float(np.median(lat)) * 1e3 if lat else None
    if stamps is not None and rank == 0:
        st = stamps.cpu().numpy().astype(np.float64)
        st = st[st[:, 13] > 0]
        why = (stamps.cpu().numpy()[:, 10] >> 32)[stamps.cpu().numpy()[:, 13] > 0]
        st[:, 10] = stamps.cpu().numpy()[:, 10][stamps.cpu().numpy()[:, 13] > 0] & 0xffffffff
        for p_ in range(int(st[:, 14].max()) + 1):
            sel = st[:, 14] == p_
            sys.stderr.write("  last solve in pass %d: %d controllers; fast path taken %.3f, multiplier <= 0 %.3f, row violated %.3f, not tried %.3f\n"
                             % (p_, sel.sum(), (why[sel] == 0).mean(), (why[sel] == 1).mean(), (why[sel] == 2).mean(), (why[sel] == 3).mean()))
        names = ["recurrences", "qd", "hessian/rows", "cholesky", "transform", "slack init", "warm start", "active-set loop", "recover z",
                 "predict"]
        seq = np.stack([st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], st[:, 5], st[:, 6], st[:, 15], st[:, 7], st[:, 8], st[:, 9]], 1)
        d = np.diff(seq, axis=1)
        sys.stderr.write("solve stamps (%d controllers, last solve each; shader clock cycles): median / mean / max\n" % len(st))
        sys.stderr.write("  %-16s %8.0f %8.0f %8.0f\n" % ("prologue", np.median(st[:, 0] - st[:, 12]), (st[:, 0] - st[:, 12]).mean(), (st[:, 0] - st[:, 12]).max()))
        for i, nm_ in enumerate(names):
            sys.stderr.write("  %-16s %8.0f %8.0f %8.0f\n" % (nm_, np.median(d[:, i]), d[:, i].mean(), d[:, i].max()))
        sys.stderr.write("  %-16s %8.0f %8.0f %8.0f\n" % ("epilogue", np.median(st[:, 13] - st[:, 9]), (st[:, 13] - st[:, 9]).mean(), (st[:, 13] - st[:, 9]).max()))
        tot_c = st[:, 13] - st[:, 12]
        sys.stderr.write("  %-16s %8.0f %8.0f %8.0f   iterations mean %.2f; clock %.2f GHz\n" % ("kernel", np.median(tot_c), tot_c.mean(), tot_c.max(), st[:, 10].mean(),
                         np.median((st[:, 9] - st[:, 0]) / np.maximum(st[:, 11], 1)) * 0.1))
    qp_fail, qp_solves, qp_iters = ctrl.stats()
    failed_steps = int((ctrl.logs["qp_status"] != 0).sum().item())      # (controller, step) pairs that held their plan
    if qp_fail > 0.01 * max(qp_solves, 1):
        raise SystemExit("%d of %d QP solves failed in the closed loop: result invalid" % (qp_fail, qp_solves))

    region_max = [shard.reduce_stats([m], "max")[0] for m in region_ms]
    elapsed_max = region_max[0] * 1e-3
    lpv = ctrl.logs["lpv_counter"][:, W:W + K]
    passes = float(lpv.sum().item())                 # controller-passes embedded in the timed region
    # final exchange: trajectories to rank 0, statistics to every rank (once per run, outside the loop)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gathered = shard.gather_logs({k: ctrl.logs[k] for k in ("u_log", "y_log", "lpv_counter")}, dst=0, sizes=sizes)
    torch.cuda.synchronize(dev)
    t_gather = time.perf_counter() - t0
    tot_passes, n_ranks_seen, tot_ctrl = shard.reduce_stats([passes, 1.0, float(B)], "sum")
    if rank != 0:
        return None
    assert gathered["y_log"].shape[0] == int(tot_ctrl)
    value = tot_ctrl * K / elapsed_max
    n_launch, ms_embed = prof["embed"]
    sampled = [k for k in range(k_inst0, k_inst0 + KI) if every and k % every == 0]
    if n_launch == 0 or not sampled:
        n_launch, ms_embed, sampled = 0, float("nan"), []
    # algorithmic flops of the event-bracketed launches: passes of this rank in the sampled steps
    fpp = NODES_EVALUATED * FLOPS_PER_POINT_NODE
    # (passes from tail_from on run in the per-controller tail launch, which is timed with the solves)
    lpv_all = ctrl.logs["lpv_counter"]
    wide = lpv_all if not ctrl.tail_from else torch.clamp(lpv_all, max=ctrl.tail_from)
    embed_points = NC * float(wide[:, sampled].sum().item()) if sampled else float("nan")
    embed_flops = fpp * embed_points
    iso, iso_all = isolated_embed_tflops(nmpc_amd, enc, lam, w, B * NC, dev)
    lpv = lpv_all[:, W:W + K]
    y = gathered["y_log"][:B, W + 1:W + K + 1]
    track = float((y - torch.clamp(torch.as_tensor(refs[:, W:W + K], device=dev), ylo, yhi)).abs().median().item())
    peak = FP16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
    achieved = embed_flops / (ms_embed * 1e-3) / 1e12
    n_solve, ms_solve = prof["solve"]
    traffic, traffic_src = pmc_traffic("r3_pmc_embed_loop.json" if NC < 24 else "r3_cfg3_pmc_embed.json")
    # algorithmic HBM bytes of a launch: 5 fp32 in, 8 fp32 out per scheduling point (x, u, xc; A, B, C) + both packed nets once
    alg_bytes = 52.0 * embed_points / max(n_launch, 1) + 2 * 21024 * 4
    roof_embed = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                  "traffic": traffic, "traffic_detail": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                  "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                  "peak_note": "dense f16 MFMA peak (16 x 157.3 at 2.4 GHz) / 3 products per f32 product; achieved = algorithmic f32 "
                               "flops of the %d nodes the kernel evaluates (the lambda = 0 node of the 41 is a host "
                               "constant) / event-timed launch span in the instrumented region; launches of different "
                               "slices overlap each other, so in-loop spans are an upper bound of the kernel's own time" % NODES_EVALUATED,
                  "clock_ghz_sustained": clock_ghz,
                  "frac_at_sustained_clock": (achieved / (peak * clock_ghz / 2.4)) if clock_ghz else None,
                  "clock_note": "median over the workgroups of the instrumented region's first-pass launches of "
                                "(shader-clock cycles) / (100 MHz real-time ticks) between workgroup start and end; peak assumes 2.4 GHz",
                  "achieved_isolated": iso, "frac_isolated": iso / peak, "achieved_isolated_runs": iso_all,
                  "isolated_note": "one full pass (%d points) alone on the GPU: median of 5 runs of 20 back-to-back launches" % (B * NC),
                  "achieved_over_wall": fpp * NC * passes / elapsed_max / 1e12,
                  "frac_of_f32_mfma_peak_isolated": iso / FP32_MFMA_PEAK_TFLOPS,
                  "kernel": "embed_kernel", "launches": n_launch, "avg_launch_ms": ms_embed / max(n_launch, 1),
                  "flops_per_launch": embed_flops / max(n_launch, 1)}
    solve_passes = float(lpv_all[:, sampled].sum().item()) if sampled else float("nan")
    if NC >= 24:
        # the long-horizon solve is dense fp64 linear algebra: priced against the fp64 rate
        f_solve, f_iter = wide_solve_flops(NC)
        its = qp_iters / max(qp_solves, 1)
        ach_s = (f_solve + its * f_iter) * solve_passes / (ms_solve * 1e-3) / 1e12 if n_solve else float("nan")
        tr_s, tr_s_src = pmc_traffic("r3_cfg3_pmc_solve.json")
        roof_solve = {"bound": "mfma", "pipe": "fp64", "achieved": ach_s, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": ach_s / FP64_PEAK_TFLOPS, "traffic": tr_s, "traffic_detail": tr_s_src,
                      "kernel": "mpc_solve_wide_kernel", "launches": n_solve, "avg_launch_ms": ms_solve / max(n_solve, 1),
                      "flops_per_solve": f_solve + its * f_iter, "flops_per_launch": (f_solve + its * f_iter) * solve_passes / max(n_solve, 1),
                      "note": "condense + QP + prediction of one controller per workgroup of four wavefronts, fp64: "
                              "Gamma recurrences, Hessian, Cholesky, transform of the constraint normals, active-set iterations "
                              "(%.2f per solve); algorithmic flops / event-timed launch span against the MI355X fp64 rate" % its}
    else:
        sb = solve_bytes_per_controller_pass(NC)
        ach_s = sb * solve_passes / (ms_solve * 1e-3) / 1e9 if n_solve else float("nan")
        tr_s, tr_s_src = pmc_traffic("r3_pmc_solve.json")
        roof_solve = {"bound": "hbm", "achieved": ach_s, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_s / HBM_PEAK_GBS,
                      "traffic": tr_s, "traffic_detail": tr_s_src, "kernel": "mpc_solve_kernel", "launches": n_solve,
                      "avg_launch_ms": ms_solve / max(n_solve, 1), "bytes_per_controller_pass": sb,
                      "note": "condense + QP + prediction, one wavefront per controller, everything between A/B/C and the "
                              "plan stays in LDS/registers: latency-bound, HBM traffic is the algorithmic minimum%s"
                              % ("; launches include the per-controller tail launch (passes >= %d: embedding + solve in one kernel)"
                                 % (ctrl.tail_from + 1) if ctrl.tail_from else "")}
    dominant_solve = n_solve and ms_solve > ms_embed
    res = {
        "metric": "closed-loop NMPC controller-steps/sec (B=4096, N=10)" if cfgname == "cfg2" else
                  "closed-loop NMPC controller-steps/sec (%s: %d controllers, N=%d)" % (cfgname, int(tot_ctrl), NC),
        "value": value, "unit": "controller-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed_max / K * 1e3, "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None,
        "dtype": "f32 via split-f16 MFMA, f32 accumulate (embedding) + f64 (condensing, QP, plant)",
        "data": "synthetic",
        "config": {"workload": "%s: unbalanced-disc encoder model (SNR30 weights), %d controllers on this rank of %d in "
                               "total, Nc=%d, stages=20 Simpson (41 nodes), max %d LPV passes tol %g, random-level "
                               "references 10-15 samples in [-1,1], u in [%g,%g], y in [%g,%g], sigma_n %g, observer in the "
                               "loop" % (cfgname, B, int(tot_ctrl), NC, cfg["max_it"], cfg["tol"], ulo, uhi, ylo, yhi, cfg["sigma"]),
                   "batch_per_gpu": B, "controllers_total": int(tot_ctrl), "Nc": NC,
                   "parallelism": "dp%d (independent shards, no collective in the loop)" % world},
        "n_ranks_seen": int(round(n_ranks_seen)),
        "ms_per_step_repeats": [m / K for m in region_max],
        "ms_per_step_spread": {"min": min(region_max) / K, "median": float(np.median(region_max)) / K, "max": max(region_max) / K,
                               "note": "%d back-to-back timed regions of %d steps each, none of them instrumented; value / "
                                       "ms_per_step are region 0" % (R, K)},
        "instrumented_region": None if not KI else {
            "steps": KI, "ms_per_step": inst_ms / KI, "event_every": every,
            "note": "run after the timed regions, not part of value: HIP event pairs on the launch streams around every "
                    "launch of every %d-th step, embedding workgroups stamp their clock; roofline figures come from here" % every},
        "qp_failed_solves": int(qp_fail), "qp_failed_controller_steps": failed_steps,
        "qp_solves": int(qp_solves), "qp_solves_by_the_full_active_set_kernel": int(getattr(ctrl, "slow_solves", 0)),
        "p50_step_latency_ms": p50_latency,
        "mean_lpv_passes_per_step": tot_passes / (tot_ctrl * K),
        "mean_qp_active_set_iterations": qp_iters / max(qp_solves, 1),
        "kernel_ms_per_step": {k: prof[k][1] / max(len(sampled), 1) for k in prof},
        "launches_per_step": sum(prof[k][0] for k in prof) / max(len(sampled), 1),
        "kernel_ms_note": "HIP event pairs on %d of the %d steps of the instrumented region; launches of different "
                          "slices of the batch run concurrently on separate streams, so these spans overlap and their sum "
                          "exceeds ms_per_step" % (len(sampled), KI),
        "target_selector_s": t_targets, "final_gather_s": t_gather,
        "median_abs_tracking_error": track,
        "roofline": roof_solve if dominant_solve else roof_embed,
        "roofline_other_kernel": roof_embed if dominant_solve else roof_solve,
    }
    return res


def wide_bench(args, rank, world, dev, shard):
    """cfg5: one open-loop LPV iteration (embedding, condensing, QP assembly, QP) of the synthetic wide
    model per step; the 65536 controllers are split over the ranks (strong scaling)."""
    import torch
    import torch.distributed as dist
    import nmpc_amd
    cfg = CONFIGS["cfg5"]
    nx, nu, ny, H, Nc = 8, 1, 1, 256, cfg["Nc"]
    total = args.batch or cfg["batch"]
    lo, hi = shard.shard_range(total, rank, world)
    B = hi - lo
    rng = np.random.default_rng(0)

    def net(n_in, n_out, lin_scale=1.0):     # weights N(0, 1/fan_in), biases N(0, 0.01); same on every rank
        g = lambda r, c: (rng.standard_normal((r, c)) / np.sqrt(c)).astype(np.float32)
        b = lambda r: (0.1 * rng.standard_normal(r)).astype(np.float32)
        return [lin_scale * g(n_out, n_in), b(n_out), g(H, n_in), b(H), g(H, H), b(H), g(n_out, H), b(n_out)]
    enc = nmpc_amd.Encoder({"fn": net(nx + nu, nx, 0.5), "hn": net(nx, ny)}, device=dev)
    g = torch.Generator().manual_seed(1000 + rank)
    x0 = torch.randn(B, nx, generator=g, dtype=torch.float64).to(dev)
    xpred = torch.randn(B, Nc, nx, generator=g, dtype=torch.float64).to(dev)
    u = (torch.rand(B, Nc, 1, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
    xs = torch.cat([x0[:, None, :], xpred[:, :-1, :]], 1)
    lam = (np.arange(20) + 1) / 20.0
    w = np.full(20, 0.05)
    Xs = torch.zeros(B, Nc * nx, dtype=torch.float64, device=dev)
    Us = torch.zeros(B, Nc, dtype=torch.float64, device=dev)
    ym = torch.zeros(B, dtype=torch.float64, device=dev)
    f0, h0 = enc.f0, enc.h0
    ev_e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    emb_ms = []

    def iteration(timed=False):
        if timed:
            ev_e[0].record()
        A, Bm, C = nmpc_amd.embed(enc, xs, u, xpred, lam, w)
        if timed:
            ev_e[1].record()
        r = nmpc_amd.mpc_qp(A.double().reshape(B, Nc * nx, nx), Bm.double().reshape(B, Nc * nx, 1), C.double().reshape(B, Nc, nx),
                            x0, ym, Xs, Us, f0, h0, np.eye(nx), 1.0, 1000.0, -1.0, 1.0, -1.0, 1.0, Nc, nx)
        if timed:
            torch.cuda.synchronize(dev)
            emb_ms.append(ev_e[0].elapsed_time(ev_e[1]))
        return r
    K, W = args.steps, args.warmup
    for _ in range(W):
        iteration()
    torch.cuda.synchronize(dev)
    region_ms = []
    for r in range(max(1, args.repeats)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(K):
            res = iteration()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        region_ms.append((time.perf_counter() - t0) * 1e3)
    for _ in range(3):
        iteration(timed=True)
    nfail = int((res["status"] != 0).sum())
    region_max = [shard.reduce_stats([m], "max")[0] for m in region_ms]
    n_ranks_seen, tot_ctrl, tot_fail = shard.reduce_stats([1.0, float(B), float(nfail)], "sum")
    gathered = shard.gather_logs({"u": res["u"]}, dst=0,
                                 sizes=[b - a for a, b in (shard.shard_range(total, r, world) for r in range(world))])
    if rank != 0:
        return None
    flops_pt = 20 * ((2 * H * (nx + nu) + 2 * H * H + 2 * nx * H * H + 2 * nx * H * (nx + nu))
                     + (2 * H * nx + 2 * H * H + 2 * ny * H * H + 2 * ny * H * nx))
    te = float(np.median(emb_ms))
    ach = B * Nc * flops_pt / (te * 1e-3) / 1e12
    f32_wide = os.environ.get("NMPC_EMBED_WIDE") == "f32"      # diagnostic switch: the exact-fp32 streaming kernel
    peak = FP32_MFMA_PEAK_TFLOPS if f32_wide else FP16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
    # HBM traffic of an embedding launch (mean over the f-net and h-net launches of the committed PMC passes); algorithmic:
    # inputs and outputs of the points (x, u / xc in, A, B / C out, fp32) + one net's fragments and tables once
    tr5, tr5_src = (None, None) if f32_wide else pmc_traffic("r3_cfg5_pmc_embed.json")
    alg5 = 0.5 * (B * Nc * 4.0 * (nx + nu + nx + nx * nx + nx * nu + ny * nx)) + 2.0 * H * H * 4 + 4.0 * H * 32
    return {
        "metric": "open-loop LPV-NMPC controller-iterations/sec (cfg5: nx=8, hidden 256, N=20, %d controllers)" % int(tot_ctrl),
        "value": tot_ctrl * K / (region_max[0] * 1e-3), "unit": "controller-iterations/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": region_max[0] / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32 MFMA (embedding) + f64 (condensing, QP)" if f32_wide else
                  "f32 via split-f16 MFMA, f32 accumulate (embedding) + f64 (condensing, QP)"), "data": "synthetic",
        "config": {"workload": "cfg5: synthetic nets nx=8 nu=ny=1 hidden 256 (weights N(0,1/fan_in), seed 0), 20 lambda nodes, "
                               "Nc=20, one open-loop iteration = embedding + condensing + QP assembly + QP on random points; "
                               "%d controllers on this rank of %d" % (B, int(tot_ctrl)),
                   "batch_per_gpu": B, "controllers_total": int(tot_ctrl), "Nc": Nc,
                   "parallelism": "dp%d (independent shards)" % world},
        "n_ranks_seen": int(round(n_ranks_seen)), "ms_per_step_repeats": [m / K for m in region_max],
        "qp_status_nonzero": int(tot_fail), "gathered_rows": int(gathered["u"].shape[0]),
        "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                     "frac": ach / peak, "traffic": tr5, "traffic_detail": tr5_src,
                     "algorithmic_bytes_per_launch": alg5, "traffic_over_algorithmic": (tr5 / alg5) if tr5 else None,
                     "kernel": "embed_wide_kernel" if f32_wide else "embed_panel_kernel",
                     "avg_launch_ms": te, "flops_per_launch": B * Nc * flops_pt,
                     "launch_note": "one embedding call = two launches (f net, h net), event-timed together",
                     "peak_note": ("exact-f32 MFMA (v_mfma_f32_16x16x4_f32) peak" if f32_wide else
                                   "dense f16 MFMA peak (16 x 157.3) / 3 products per f32 product; achieved = algorithmic f32 flops")},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0,
                    help="controllers per GPU (weak-scaling configs) or in total (strong-scaling configs); 0 = the config's")
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed regions of K steps run back to back (value = the first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: record no HIP event pairs (roofline fields become null)")
    ap.add_argument("--event-every", type=int, default=EVENT_EVERY,
                    help="bracket the kernel launches of every n-th step of the instrumented region with HIP event pairs")
    ap.add_argument("--instrumented-steps", type=int, default=INSTRUMENTED_STEPS,
                    help="steps of the instrumented region that follows the timed regions (at most --steps; 0: none)")
    ap.add_argument("--cpu-controllers", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=12, help="steps per controller of the 1-thread slice baseline")
    ap.add_argument("--cpu-steps-mt", type=int, default=120, help="steps per controller of the all-threads slice baseline")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--solve-stamps", action="store_true", help="diagnostics: per-phase clock stamps of the long-horizon solve kernel (stderr)")
    ap.add_argument("--latency-steps", type=int, default=40,
                    help="single steps run after the timed regions for p50_step_latency_ms (0: skip)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="diagnostic: run the N-rank path with every rank on cuda:0 (gloo rendezvous; two ranks cannot share "
                         "one GPU under RCCL)")
    ap.add_argument("--collectives-at-world-1", action="store_true",
                    help="diagnostic: initialise backend nccl and run the final gather / all-reduce through it even with one rank")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_main(args)
        return
    defaults = {"cfg2": (200, 20), "cfg3": (40, 10), "cfg4": (100, 20), "cfg5": (3, 1)}[args.config]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent only starts and reaps the ranks: it never imports torch or touches a GPU
        argv = [a for a in sys.argv[1:]]
        sys.exit(spawn_ranks(args.gpus, argv))

    # host baseline first, in a child that never touches the GPU (single-GPU, headline configuration only)
    rank_env = int(os.environ.get("RANK", 0))
    cpu = None
    if not args.no_cpu_baseline and args.gpus == 1 and rank_env == 0 and args.config == "cfg2":
        cpu = run_cpu_baseline(args)

    import torch
    import torch.distributed as dist
    import nmpc_amd  # noqa: F401
    from nmpc_amd import shard

    rank, world, local_rank = shard.env_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.collectives_at_world_1:
        if args.collectives_at_world_1 and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            shard.FORCE_COLLECTIVES = True
        elif args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.config == "cfg5":
        res = wide_bench(args, rank, world, dev, shard)
    else:
        res = closed_loop_bench(args, args.config, rank, world, dev, shard)
    if rank == 0:
        if dist.is_initialized():
            res["collective_backend"] = dist.get_backend()
        if cpu is not None:
            one, many = cpu["cfg2_slice_1thread"], cpu.get("cfg2_slice_nthreads")
            best = many if many and many["controller_steps_per_s"] > one["controller_steps_per_s"] else one
            res["cpu_baseline"] = {
                "value": best["controller_steps_per_s"], "unit": "controller-steps/s",
                "cores": cpu["cores"] if best is many else 1, "kind": "port",
                "sample": "fp64 NumPy oracle (restatement of the reference's CPU path), " + best.get("sample", ""),
                "host_cpus": cpu["host_cpus"], "threads_used": cpu["cores"],
                "one_thread": one, "all_threads": many,
                "cfg1_one_thread": cpu["cfg1_1thread"], "cfg1_one_thread_vmap_jacrev": cpu["cfg1_1thread_vmap_jacrev"],
                "note": "cfg1 = the reference's own signal randomLevelTime10_15Range-1_1Nsim200, one controller, 190 steps; "
                        "vmap_jacrev = Jacobians by torch.func.vmap(jacrev) as NMPC_test.py:137-177, otherwise analytic; "
                        "buckets as the reference's components_time; targets precomputed, untimed",
            }
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
