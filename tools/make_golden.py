#!/usr/bin/env python3
"""Generate golden input/output vectors by running the reference's own pure-NumPy functions.

Run ONLY in the build container (needs /root/reference); the outputs under tests/golden/ are
committed and are what travels.  Two of the reference's modules import third-party packages
that are absent here (casadi, qpsolvers) at module top although the functions exercised below
never touch them (mpcUtil.py:3-4, lpv_int.py:2).  They are imported with two EMPTY module
objects registered under those names; no function of either package is emulated.

Produces
  tests/golden/ref_condense.npz  getPhi/getGamma/getF0/getZ/getPsi/getOmega/getDEMc/getDEMc_out
                                 on seeded random A_k,B_k,C_k for several (Nc,nx,nu,ny)
  tests/golden/ref_lpv.npz       lpv_int/lpv_int_C/lambda_simpson/lambda_trap/lpv_rk4 driven with
                                 Jacobian callables of the shipped disc model, where the
                                 callables are torch.func jacrev of torch nets rebuilt from the
                                 raw weights (the mechanism of NMPC_test.py:137-138), fp64
  tests/golden/ref_jac.npz       vmap(jacrev) Jacobians of f, h at seeded points (fp64 torch)
  tests/golden/ref_ablist.npz    getABlist/getABClist/getClist (mpcUtil.py:136-165) driven with NumPy
                                 stand-ins of the horizon-mapped Get_A/Get_B/Get_C (lambda_simpson of the
                                 reference on the jacrev callables, Nc results side by side)
(`--only ablist` writes just the last file.)

`--only records` copies the data files of the reference that the tests use as they are (reference
signals and the logs of its own recorded runs), under a `ref_` prefix:
  references/randomLevelTime10_15Range-1_1Nsim200.npy, randomLevelTime25_30Range-1_1Nsim500.npy,
  randomLevelTime25_30Range-1_1Nsim5000.npy; Experiments/log_q.npy, log_w.npy, lpv_counter.npy,
  lpv_counter_tight.npy, lpv_counter_noise.npy, components_time.npy
"""
import os
import sys
import types

import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    for name in ("casadi", "qpsolvers"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import mpcUtil, lpv_int, lpv_rk4  # noqa
    sys.path.pop(0)
    return mpcUtil, lpv_int, lpv_rk4


def torch_nets(weights):
    import torch
    d = np.load(weights)

    def make(net):
        p = {k: torch.tensor(np.asarray(d["%s_%s" % (net, k)], np.float64)) for k in
             ("lin_w", "lin_b", "w0", "b0", "w2", "b2", "w4", "b4")}

        def fwd(z):
            t1 = torch.tanh(p["w0"] @ z + p["b0"])
            t2 = torch.tanh(p["w2"] @ t1 + p["b2"])
            return p["lin_w"] @ z + p["lin_b"] + p["w4"] @ t2 + p["b4"]
        return fwd
    return make("fn"), make("hn")


def make_ablist(mpcUtil, lpv_int):
    import torch
    from torch.func import jacrev
    fnet, hnet = torch_nets(os.path.join(ROOT, "nmpc_amd", "data", "disc_snr30.npz"))
    nx, nu, ny, Nc, stages = 2, 1, 1, 10, 20
    f_xu = lambda x, u: fnet(torch.cat([x, u]))
    t = lambda v, n: torch.as_tensor(np.asarray(v, np.float64)).reshape(n)
    Jfx = lambda x, u: jacrev(f_xu, argnums=0)(t(x, nx), t(u, nu)).numpy()
    Jfu = lambda x, u: jacrev(f_xu, argnums=1)(t(x, nx), t(u, nu)).numpy()
    Jhx = lambda x, u: jacrev(hnet)(t(x, nx)).numpy()

    def mapped(k):        # what Function(...).map(Nc) returns: the Nc results side by side
        def call(X, u):
            u = np.asarray(u, np.float64).reshape(-1)
            return np.hstack([lpv_int.lambda_simpson(X[:, i], u[i], nx, nu, ny, Jfx, Jfu, Jhx, stages)[k]
                              for i in range(X.shape[1])])
        return call
    rng = np.random.default_rng(777)
    B = 3
    out = {"x": rng.uniform(-2, 2, (B, Nc * nx)), "u": rng.uniform(-2, 2, (B, Nc))}   # 1-D stacked states, as np.split needs
    lA, lB, lC, lC2 = [], [], [], []
    for b in range(B):
        a, bb = mpcUtil.getABlist(out["x"][b], out["u"][b], Nc, nx, nu, mapped(0), mapped(1),
                                  np.zeros((Nc * nx, nx)), np.zeros((Nc * nx, nu)))
        c = mpcUtil.getClist(out["x"][b], out["u"][b], Nc, nx, ny, mapped(2), np.zeros((Nc * ny, nx)))
        _, _, c2 = mpcUtil.getABClist(out["x"][b], out["u"][b], Nc, nx, nu, ny, mapped(0), mapped(1), mapped(2),
                                      np.zeros((Nc * nx, nx)), np.zeros((Nc * nx, nu)), np.full((Nc * ny, nx), 7.0))
        lA.append(a); lB.append(bb); lC.append(c); lC2.append(c2)
    out.update(list_A=np.array(lA), list_B=np.array(lB), list_C=np.array(lC), list_C_abc=np.array(lC2))
    np.savez_compressed(os.path.join(OUT, "ref_ablist.npz"), **out)


RECORDS = ["references/randomLevelTime10_15Range-1_1Nsim200.npy", "references/randomLevelTime25_30Range-1_1Nsim500.npy",
           "references/randomLevelTime25_30Range-1_1Nsim5000.npy", "Experiments/log_q.npy", "Experiments/log_w.npy",
           "Experiments/lpv_counter.npy", "Experiments/lpv_counter_tight.npy", "Experiments/lpv_counter_noise.npy",
           "Experiments/components_time.npy"]


def copy_records():
    """Data files of the reference, loaded as plain arrays (no pickle) and re-saved under tests/golden/."""
    os.makedirs(OUT, exist_ok=True)
    for rel in RECORDS:
        a = np.load(os.path.join(REF, rel), allow_pickle=False)
        np.save(os.path.join(OUT, "ref_" + os.path.basename(rel)), a)
        print("copied", rel, a.shape, a.dtype)


def main():
    if sys.argv[1:3] == ["--only", "records"]:
        copy_records()
        return
    import torch
    from torch.func import jacrev, vmap
    mpcUtil, lpv_int, lpv_rk4 = load_reference()
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:3] == ["--only", "ablist"]:
        make_ablist(mpcUtil, lpv_int)
        print("wrote ref_ablist.npz")
        return
    make_ablist(mpcUtil, lpv_int)
    rng = np.random.default_rng(20240607)

    # ---- condensing ------------------------------------------------------------------
    out = {}
    cases = [(4, 2, 1, 1), (10, 2, 1, 1), (50, 2, 1, 1), (7, 3, 1, 1), (20, 8, 1, 1)]
    out["cases"] = np.array(cases)
    for ci, (Nc, nx, nu, ny) in enumerate(cases):
        lA = rng.uniform(-1.2, 1.2, (Nc * nx, nx)) * (0.9 if Nc > 20 else 1.0)
        lB = rng.uniform(-1, 1, (Nc * nx, nu))
        lC = rng.uniform(-3, 3, (Nc * ny, nx))
        f0 = rng.uniform(-0.2, 0.2, (nx, 1))
        Q = np.diag(rng.uniform(1, 1000, nx)); R = np.array([[rng.uniform(0.5, 2)]])
        lo = -rng.uniform(1, 8, nx); hi = rng.uniform(1, 8, nx)
        ylo = -rng.uniform(1, 2, ny); yhi = rng.uniform(1, 2, ny)
        umin, umax = -float(rng.uniform(2, 6)), float(rng.uniform(2, 6))
        k = "c%d_" % ci
        out[k + "list_A"], out[k + "list_B"], out[k + "list_C"], out[k + "f0"] = lA, lB, lC, f0
        out[k + "Q"], out[k + "R"] = Q, R
        out[k + "bounds"] = np.concatenate([lo, hi, ylo, yhi, [umin, umax]])
        out[k + "Phi"] = mpcUtil.getPhi(lA, Nc, nx, nu)
        out[k + "Gamma"] = mpcUtil.getGamma(lA, lB, Nc, nx, nu)
        out[k + "F0"] = mpcUtil.getF0(lA, f0, Nc, nx)
        out[k + "Z"] = mpcUtil.getZ(lC, Nc, ny, nx)
        out[k + "Psi"] = mpcUtil.getPsi(Nc, R)
        out[k + "Omega"] = mpcUtil.getOmega(Nc, Q)
        for nm, v in zip("DEMc", mpcUtil.getDEMc(list(lo), list(hi), umin, umax, Nc, nx, nu)):
            out[k + "st_" + nm] = v
        for nm, v in zip("DEMc", mpcUtil.getDEMc_out(list(ylo), list(yhi), umin, umax, Nc, ny, nu)):
            out[k + "out_" + nm] = v
    np.savez_compressed(os.path.join(OUT, "ref_condense.npz"), **out)

    # ---- Jacobians + quadrature ------------------------------------------------------
    fnet, hnet = torch_nets(os.path.join(ROOT, "nmpc_amd", "data", "disc_snr30.npz"))
    nx, nu, ny = 2, 1, 1
    f_xu = lambda x, u: fnet(torch.cat([x, u]))
    JF = vmap(jacrev(f_xu, argnums=(0, 1)))
    JH = vmap(jacrev(hnet))
    P = 256
    xs = rng.uniform(-3, 3, (P, nx)); us = rng.uniform(-3, 3, (P, nu))
    xs[0] = 0; us[0] = 0
    xs[1] *= 5; us[1] *= 3                      # saturated regime
    jfx, jfu = JF(torch.tensor(xs), torch.tensor(us))
    jhx = JH(torch.tensor(xs))
    np.savez_compressed(os.path.join(OUT, "ref_jac.npz"), x=xs, u=us, Jfx=jfx.numpy(), Jfu=jfu.numpy(),
                        Jhx=jhx.numpy())

    Jfx = lambda x, u: jacrev(f_xu, argnums=0)(torch.as_tensor(x, dtype=torch.float64).reshape(nx),
                                               torch.as_tensor(u, dtype=torch.float64).reshape(nu)).numpy()
    Jfu = lambda x, u: jacrev(f_xu, argnums=1)(torch.as_tensor(x, dtype=torch.float64).reshape(nx),
                                               torch.as_tensor(u, dtype=torch.float64).reshape(nu)).numpy()
    Jhx = lambda x, u: jacrev(hnet)(torch.as_tensor(x, dtype=torch.float64).reshape(nx)).numpy()
    out = {}
    Pq = 24
    xq = rng.uniform(-2.5, 2.5, (Pq, nx)); uq = rng.uniform(-2.5, 2.5, (Pq, nu))
    xq[0] = 0; uq[0] = 0
    out["x"], out["u"] = xq, uq
    res = {k: [] for k in ("int_A", "int_B", "intC_A", "intC_B", "intC_C", "simp_A", "simp_B", "simp_C",
                           "trap_A", "trap_B", "trap_C", "odd_A", "odd_B")}
    for x, u in zip(xq, uq):
        a, b = lpv_int.lpv_int(x, nx, u, nu, Jfx, Jfu, 0.05, 20)
        res["int_A"].append(a); res["int_B"].append(b)
        a, b, c = lpv_int.lpv_int_C(x, nx, u, nu, ny, Jfx, Jfu, Jhx, 0.05, 20)
        res["intC_A"].append(a); res["intC_B"].append(b); res["intC_C"].append(c)
        a, b, c = lpv_int.lambda_simpson(x, u, nx, nu, ny, Jfx, Jfu, Jhx, 20)
        res["simp_A"].append(a); res["simp_B"].append(b); res["simp_C"].append(c)
        a, b, c = lpv_int.lambda_trap(x, u, nx, nu, ny, Jfx, Jfu, Jhx, 20)
        res["trap_A"].append(a); res["trap_B"].append(b); res["trap_C"].append(c)
        a, b = lpv_int.lpv_int(x, nx, u, nu, Jfx, Jfu, 0.1, 7)     # dlam*stages != 1
        res["odd_A"].append(a); res["odd_B"].append(b)
    for k, v in res.items():
        out[k] = np.array(v)
    # lpv_rk4 with the embedded matrices
    dt = 0.1
    c = rng.uniform(-0.1, 0.1, nx)
    out["rk4_c"] = c
    out["rk4_next"] = np.array([lpv_rk4.lpv_rk4(x, u, a, b, dt, c) for x, u, a, b in
                                zip(xq, uq, out["int_A"], out["int_B"])])
    np.savez_compressed(os.path.join(OUT, "ref_lpv.npz"), **out)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
