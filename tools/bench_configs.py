#!/usr/bin/env python3
"""Timings of the configurations other than the headline one (bench.py covers that): one JSON line each.

  python tools/bench_configs.py [cfg3] [cfg4] [cfg5] [nc20]          (default: all)

cfg3  long horizon: Nc = 50, u in [-3, 3], y in [-1, 1] (NMPC_output.py:226-227), 4096 controllers
cfg4  noisy measurement (sigma 0.014 = SNR 30), LPV passes iterated to convergence (tol 1e-5, cap 10),
      2048 controllers (one GPU's share of 16384)
cfg5  synthetic wide model (nx = 8, hidden 256, 20 lambda-nodes, Nc = 20): one open-loop LPV iteration
      (embedding, condensing, QP) of 65536 controllers (--cfg5-batch to change)
nc20  the disc model with Nc = 20

Inputs are synthetic and seeded like bench.py's; the numbers are throughput only (parity of these
configurations is in tests/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def closed_loop(name, B, Nc, Nsim, ubox, ybox, sigma, max_it, tol, warm):
    import torch
    import nmpc_amd
    enc = nmpc_amd.load_system()
    nm = enc.norm
    refs = np.stack([nmpc_amd.randomLevelReference(Nsim + Nc, [10, 15], [-1, 1], np.random.default_rng(1234 + b))[:Nsim + Nc]
                     for b in range(B)])
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    ymin_n, ymax_n = (ybox[0] - nm.y0) / nm.ystd, (ybox[1] - nm.y0) / nm.ystd
    umin_n, umax_n = (ubox[0] - nm.u0) / nm.ustd, (ubox[1] - nm.u0) / nm.ustd
    yrn = torch.clamp((torch.as_tensor(refs, device="cuda") - nm.y0) / nm.ystd, ymin_n, ymax_n)
    Xs, Us, _ = nmpc_amd.getXsUs_Cs(yrn, 2, 1, 1, Nsim + Nc, umin_n, umax_n, [ymin_n], [ymax_n], nmpc_amd.GetABC(enc, lam, w))
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[sigma])
    ctrl = nmpc_amd.Controller(enc, plant, B, Nc, Nsim, max_it, np.diag([1000.0, 10.0]), 1.0, umin_n, umax_n, ymin_n, ymax_n,
                               lam, w, tol=tol, rho=1000.0)
    noise = None
    if sigma > 0:
        noise = torch.as_tensor(sigma * np.random.default_rng(99).standard_normal((Nsim, B)), device="cuda")
    ctrl.set_targets(Xs.transpose(1, 2).contiguous(), Us[:, 0, :].contiguous(), noise)
    ctrl.reset((np.zeros(2) - nm.y0) / nm.ystd, (0 - nm.u0) / nm.ustd)
    ctrl.run(0, warm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ctrl.run(warm, Nsim - warm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    K = Nsim - warm
    qp_fail, solves, iters = ctrl.stats()
    y = ctrl.logs["y_log"][:, warm + 1:Nsim + 1].cpu().numpy()
    return {"config": name, "controllers": B, "Nc": Nc, "steps": K, "ms_per_step": dt / K * 1e3,
            "controller_steps_per_s": B * K / dt, "mean_lpv_passes": float(ctrl.logs["lpv_counter"][:, warm:Nsim].float().mean()),
            "qp_failures": int(qp_fail), "mean_qp_iterations": iters / max(solves, 1),
            "median_abs_tracking_error": float(np.median(np.abs(y - np.clip(refs[:, warm:Nsim], ybox[0], ybox[1]))))}


def wide_open_loop(B):
    import torch
    import nmpc_amd
    nx, nu, ny, H, Nc = 8, 1, 1, 256, 20
    rng = np.random.default_rng(0)

    def net(n_in, n_out, lin_scale=1.0):     # weights N(0, 1/fan_in), biases N(0, 0.01)
        g = lambda r, c: (rng.standard_normal((r, c)) / np.sqrt(c)).astype(np.float32)
        b = lambda r: (0.1 * rng.standard_normal(r)).astype(np.float32)
        return [lin_scale * g(n_out, n_in), b(n_out), g(H, n_in), b(H), g(H, H), b(H), g(n_out, H), b(n_out)]
    enc = nmpc_amd.Encoder({"fn": net(nx + nu, nx, 0.5), "hn": net(nx, ny)}, device="cuda:0")
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(B, nx, generator=g, dtype=torch.float64).cuda()
    xpred = torch.randn(B, Nc, nx, generator=g, dtype=torch.float64).cuda()
    u = (torch.rand(B, Nc, 1, generator=g, dtype=torch.float64) * 2 - 1).cuda()
    xs = torch.cat([x0[:, None, :], xpred[:, :-1, :]], 1)
    lam = (np.arange(20) + 1) / 20.0
    w = np.full(20, 0.05)
    Xs = torch.zeros(B, Nc * nx, dtype=torch.float64).cuda()
    Us = torch.zeros(B, Nc, dtype=torch.float64).cuda()
    ym = torch.zeros(B, dtype=torch.float64).cuda()
    f0, h0 = enc.f0, enc.h0

    def iteration():
        A, Bm, C = nmpc_amd.embed(enc, xs, u, xpred, lam, w)
        return nmpc_amd.mpc_qp(A.double().reshape(B, Nc * nx, nx), Bm.double().reshape(B, Nc * nx, 1), C.double().reshape(B, Nc, nx),
                               x0, ym, Xs, Us, f0, h0, np.eye(nx), 1.0, 1000.0, -1.0, 1.0, -1.0, 1.0, Nc, nx)
    iteration()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    nmpc_amd.embed(enc, xs, u, xpred, lam, w)
    ev[1].record()
    r = iteration()
    ev[2].record()
    torch.cuda.synchronize()
    te, tt = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    flops = B * Nc * 20 * ((2 * H * (nx + nu) + 2 * H * H + 2 * nx * H * H + 2 * nx * H * (nx + nu))
                           + (2 * H * nx + 2 * H * H + 2 * ny * H * H + 2 * ny * H * nx))
    return {"config": "cfg5", "controllers": B, "Nc": Nc, "embed_ms": te, "embed_tflops": flops / te / 1e9,
            "iteration_ms": tt, "controller_iterations_per_s": B / tt * 1e3, "qp_status_nonzero": int((r["status"] != 0).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["cfg3", "cfg4", "cfg5", "nc20"])
    ap.add_argument("--cfg5-batch", type=int, default=65536)
    args = ap.parse_args()
    for c in args.configs:
        if c == "cfg3":
            r = closed_loop("cfg3", 4096, 50, 40, (-3, 3), (-1.0, 1.0), 0.0, 4, 1e-1, 10)
        elif c == "cfg4":
            r = closed_loop("cfg4", 2048, 10, 120, (-4, 4), (-1.2, 1.2), 0.014, 10, 1e-5, 20)
        elif c == "nc20":
            r = closed_loop("nc20", 4096, 20, 60, (-4, 4), (-1.2, 1.2), 0.0, 4, 1e-1, 10)
        elif c == "cfg5":
            r = wide_open_loop(args.cfg5_batch)
        else:
            raise SystemExit("unknown configuration %r" % c)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
