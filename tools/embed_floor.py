#!/usr/bin/env python3
"""Embedding-only measurements on one GPU (the figures DESIGN.md quotes for the launch-size dependence of the embedding
kernel and for the embedding floor of a closed-loop step).

  python tools/embed_floor.py sizes            one full pass alone on the GPU at 13 650 / 40 960 / 204 800 points and at
                                               40 / 80 / 160 nodes (median of 5 x 20 back-to-back launches)
  python tools/embed_floor.py step             the embedding launches of one closed-loop step (a first pass and a second
                                               pass of about half the controllers per slice) on 1..3 streams, issued back
                                               to back and replayed from a HIP graph (no host time in the figure)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmpc_amd  # noqa: E402

FLOPS_PER_POINT_NODE = 42624          # f net 25 728 + h net 16 896 (hidden 64, nx 2, nu 1, ny 1)
PEAK = 16 * 157.3 / 3.0               # split-fp16 ceiling, TFLOP/s
dev = "cuda:0"


def bufs(n):
    g = torch.Generator(device="cpu").manual_seed(n)
    x = torch.randn(n, 2, generator=g).to(dev).float()
    u = (torch.rand(n, 1, generator=g) * 2 - 1).to(dev).float()
    xc = torch.randn(n, 2, generator=g).to(dev).float()
    return x, u, xc, (torch.empty(n, 2, 2, device=dev), torch.empty(n, 2, 1, device=dev), torch.empty(n, 1, 2, device=dev))


def sizes(enc):
    for npts, nn in ((13650, 40), (40960, 40), (204800, 40), (40960, 80), (40960, 160)):
        if nn == 40:
            lam, w = nmpc_amd.quad_nodes("simpson", 20)
            evaluated = 40
        else:
            lam = (np.arange(nn) + 1.0) / nn; w = np.full(nn, 1.0 / nn); evaluated = nn
        x, u, xc, out = bufs(npts)
        for _ in range(5):
            nmpc_amd.embed(enc, x, u, xc, lam, w, out=out)
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                nmpc_amd.embed(enc, x, u, xc, lam, w, out=out)
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 20 * 1e3)
        us = float(np.median(res))
        tf = evaluated * FLOPS_PER_POINT_NODE * npts / us / 1e6
        print("%7d points, %3d nodes: %7.1f us per pass, %.2f ns per point, %.1f TFLOP/s = %.3f of the split ceiling"
              % (npts, evaluated, us, us * 1e3 / npts, tf, tf / PEAK))


def step(enc):
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    rep = 20
    for launch_sizes, nstreams in (((40960, 21300), 1), ((20480, 10650), 2), ((13650, 7100), 3)):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        data = [[bufs(n) for n in launch_sizes] for _ in range(nstreams)]

        def steps():
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            for _ in range(rep):
                for s, d in zip(streams, data):
                    with torch.cuda.stream(s):
                        for (x, u, xc, out) in d:
                            nmpc_amd.embed(enc, x, u, xc, lam, w, out=out)
            for s in streams:
                cur.wait_stream(s)
        def direct():
            for s, d in zip(streams, data):
                with torch.cuda.stream(s):
                    for (x, u, xc, out) in d:
                        nmpc_amd.embed(enc, x, u, xc, lam, w, out=out)
        for _ in range(10):
            direct()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            direct()
        torch.cuda.synchronize()
        ms_direct = (time.perf_counter() - t0) / 200 * 1e3
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            steps()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            steps()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        ms_graph = (time.perf_counter() - t0) / (10 * rep) * 1e3
        del g
        torch.cuda.synchronize()
        pts = nstreams * sum(launch_sizes)
        print("%d stream(s) x launches of %s points (%d per step): %.4f ms per step issued from Python (host time included), %.4f ms replayed "
              "from a graph (%.2f ns per point)" % (nstreams, launch_sizes, pts, ms_direct, ms_graph, ms_graph * 1e6 / pts))


if __name__ == "__main__":
    enc = nmpc_amd.load_system(device=dev)
    what = sys.argv[1] if len(sys.argv) > 1 else "sizes"
    (sizes if what == "sizes" else step)(enc)
