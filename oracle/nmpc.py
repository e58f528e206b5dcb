"""fp64 restatement of the reference's closed-loop LPV-NMPC controllers.

`output_nmpc_linear` follows NMPC_test.py:22-320 (output + input constraints through
Z = blkdiag(C_k), observer in the loop), `target_selector` follows mpcUtil.py:241-320
(getXsUs_Cs).  Where the reference's code departs from its own evident intent the INTENDED
math is used, and where a quirk is a deliberate property of the reference's matrices it is
kept:
  kept     - Gamma/F0 index pattern (see oracle.mpcutil);
           - x0 is normalised with the OUTPUT scaling (NMPC_test.py:48);
           - one scalar slack on every constraint row, input rows included, free sign
             (NMPC_test.py:223-226);
           - target-selector inequality right-hand side without the h0 shift: the
             (4,1)-(4,) broadcast at mpcUtil.py:254 followed by h[:,0] at :302 drops it.
  intended - scheduling points (lam x_k, lam u_k) for A_k,B_k and lam x_{k+1} for C_k instead
             of the kron/reshape pairing of NMPC_test.py:167-173;
           - h = c - D y - M Z (Phi x0 + F0) - M H0 as a vector (NMPC_test.py:217 broadcasts
             it to a matrix).
"""
import numpy as np

from . import lpv, mpcutil
from .plant import disc_step
from .qp import solve_qp


def target_selector(model, y_ref_norm, Nsim, u_min, u_max, y_min, y_max, nodes, weights,
                    max_lpv=20, tol=1e-4, lam=1000.0):
    """Steady-state targets (xs, us) per reference sample, warm-started along the sequence."""
    nx, nu, ny, ne = model.nx, model.nu, model.ny, 1
    assert nu == 1 and ny == 1
    Q = np.eye(ny); R = np.eye(nu)
    Mi = np.vstack([np.zeros((nu, ny)), np.zeros((nu, ny)), -np.eye(ny), np.eye(ny)])
    Ei = np.vstack([-np.eye(nu), np.eye(nu), np.zeros((ny, nu)), np.zeros((ny, nu))])
    hvec = np.concatenate([[-u_min, u_max], -np.atleast_1d(y_min), np.atleast_1d(y_max)]).astype(float)
    f0 = np.asarray(model.f0).reshape(nx); h0 = np.asarray(model.h0).reshape(ny)
    nv = nx + nu + ne
    T = np.zeros((2 * (ny + nu), nv)); T[:, nx:nx + nu] = Ei; T[:, nx + nu:] = -1.0
    P = np.zeros((nv, nv)); P[nx:nx + nu, nx:nx + nu] = R; P[nx + nu:, nx + nu:] = lam
    q = np.zeros(nv)
    Aeq = np.zeros((nx + ny, nv)); beq = np.zeros(nx + ny); beq[:nx] = f0
    xs = np.zeros(nx); us = np.zeros(nu); e = np.zeros(ne)
    Xs = np.zeros((nx, Nsim)); Us = np.zeros((nu, Nsim)); Es = np.zeros((ne, Nsim))
    iters = np.zeros(Nsim, int)
    for j in range(Nsim):
        beq[nx:] = y_ref_norm[j] - h0
        for i in range(max_lpv):
            As, Bs, Cs = lpv.embed(model, xs, us, nodes, weights)
            T[:, :nx] = Mi @ Cs
            Aeq[:nx, :nx] = np.eye(nx) - As
            Aeq[:nx, nx:nx + nu] = -Bs
            Aeq[nx:, :nx] = Cs
            q[:nx] = Cs.T @ Q @ (h0 - y_ref_norm[j])
            P[:nx, :nx] = Cs.T @ Q @ Cs
            z = solve_qp(P, q, T, hvec, Aeq, beq)
            xold, uold = xs, us
            xs, us, e = z[:nx], z[nx:nx + nu], z[nx + nu:]
            iters[j] = i + 1
            if np.linalg.norm(xs - xold) <= tol and np.linalg.norm(us - uold) <= tol:
                break
        Xs[:, j], Us[:, j], Es[:, j] = xs, us, e
    return Xs, Us, Es, iters


def mpc_qp(list_A, list_B, list_C, x0, y_meas, Xs, Us, f0, h0, Psi, Omega, D, E, M, c, Nc, nx, nu, ny,
           rho):
    """One condensed QP of the output-constrained controller (NMPC_test.py:206-238).
    Returns (u_plan, eps, x_pred, pieces) with pieces = dict of the assembled matrices."""
    F0 = mpcutil.getF0(list_A, f0, Nc, nx)[:, 0]
    Phi = mpcutil.getPhi(list_A, Nc, nx, nu)
    Gamma = mpcutil.getGamma(list_A, list_B, Nc, nx, nu)
    Z = mpcutil.getZ(list_C, Nc, ny, nx)
    G = 2 * (Psi + Gamma.T @ Omega @ Gamma)
    F = 2 * (Gamma.T @ Omega @ (Phi @ x0 - Xs) - Psi @ Us + Gamma.T @ Omega @ F0)
    L = M @ Z @ Gamma + E
    H0 = np.tile(np.asarray(h0).reshape(ny), Nc)
    hv = c[:, 0] - D @ np.atleast_1d(y_meas) - M @ (Z @ (Phi @ x0 + F0)) - M @ H0
    nvar = Nc * nu + 1
    Ge = np.zeros((nvar, nvar)); Ge[:-1, :-1] = G; Ge[-1, -1] = rho
    Fe = np.append(F, 0.0)
    Le = np.hstack([L, -np.ones((L.shape[0], 1))])
    ue = solve_qp(Ge, Fe, Le, hv)
    u = ue[:-1]
    xpred = Phi @ x0 + Gamma @ u + F0
    return u, ue[-1], xpred, dict(Phi=Phi, Gamma=Gamma, F0=F0, Z=Z, G=G, F=F, L=L, h=hv, Ge=Ge, Fe=Fe, Le=Le)


def output_nmpc_linear(model, u_min, u_max, y_min, y_max, x0, Q, R, dt, stages, y_reference_list,
                       Nc=5, Nsim=30, max_iterations=1, tol=1e-1, rho=1000.0, noise=None,
                       rule="simpson", dlam=None, targets=None, plant_x0=None, merge=True):
    """Closed loop of the output-constrained LPV-NMPC with observer, single controller.

    noise: optional array [Nsim] added to the measured angle (sigma_n of the plant).
    Returns a dict of logs named as the reference's return tuple (NMPC_test.py:320).
    """
    nx, nu, ny = model.nx, model.nu, model.ny
    nodes, weights = lpv.quad_nodes(rule, stages, dlam)
    if merge:
        nodes, weights = lpv.merge_nodes(nodes, weights)
    y_ref_n = (np.asarray(y_reference_list, float) - model.y0) / model.ystd
    x0 = np.asarray(x0, float)
    x0n = (x0 - model.y0) / model.ystd
    u0n = (0 - model.u0) / model.ustd
    ymax_n = (np.atleast_1d(y_max).astype(float) - model.y0) / model.ystd
    ymin_n = (np.atleast_1d(y_min).astype(float) - model.y0) / model.ystd
    umin_n = (u_min - model.u0) / model.ustd
    umax_n = (u_max - model.u0) / model.ustd
    y_ref_n = np.clip(y_ref_n, ymin_n, ymax_n)
    if targets is None:
        Xs_all, Us_all, _, _ = target_selector(model, y_ref_n, Nsim + Nc, umin_n, umax_n, ymin_n, ymax_n,
                                               nodes, weights)
    else:
        Xs_all, Us_all = targets
    f0 = np.asarray(model.f0).reshape(nx); h0 = np.asarray(model.h0).reshape(ny)
    Psi = mpcutil.getPsi(Nc, R); Omega = mpcutil.getOmega(Nc, Q)
    D, E, M, c = mpcutil.getDEMc_out(ymin_n, ymax_n, umin_n, umax_n, Nc, ny, nu)

    x = np.tile(x0n, Nc)           # predicted states x_1..x_Nc
    u = np.tile(u0n, Nc).astype(float)
    uhist = np.zeros(model.na); yhist = np.zeros(model.nb + 1)
    y_meas_n = np.zeros(ny)
    xp = np.zeros(2) if plant_x0 is None else np.asarray(plant_x0, float).copy()
    x0n = x0n.copy()

    u_log = np.zeros(Nsim); x_log = np.zeros((Nsim + 1, nx)); x_log[0] = x0
    y_log = np.zeros(Nsim + 1); y_est_log = np.zeros(Nsim + 1); e_log = np.zeros(Nsim)
    lpv_counter = np.zeros(Nsim, int)
    plan_log = np.zeros((Nsim, Nc)); abc_first = None
    for k in range(Nsim):
        Xs = Xs_all[:, k + 1:k + Nc + 1].T.reshape(nx * Nc)
        Us = Us_all[:, k:k + Nc].T.reshape(Nc * nu)
        while True:
            xsched = np.concatenate([x0n, x[:-nx]]).reshape(Nc, nx)
            A, B, C = lpv.embed(model, xsched, u.reshape(Nc, nu), nodes, weights, xc=x.reshape(Nc, nx))
            lA, lB, lC = A.reshape(Nc * nx, nx), B.reshape(Nc * nx, nu), C.reshape(Nc * ny, nx)
            if abc_first is None:
                abc_first = (lA.copy(), lB.copy(), lC.copy())
            u_old = u.copy()
            u, eps, x, _ = mpc_qp(lA, lB, lC, x0n, y_meas_n, Xs, Us, f0, h0, Psi, Omega, D, E, M, c,
                                  Nc, nx, nu, ny, rho)
            lpv_counter[k] += 1
            if lpv_counter[k] >= max_iterations or np.linalg.norm(u - u_old) < tol:
                break
        plan_log[k] = u
        u_den = model.ustd * u[0] + model.u0
        xp = disc_step(xp, u_den, dt)
        y_meas = xp[1] + (0.0 if noise is None else noise[k])
        y_meas_n = np.atleast_1d((y_meas - model.y0) / model.ystd)
        uhist = np.append(uhist[1:], u[0])
        yhist = np.append(yhist[1:], y_meas_n[0])
        zest = model.encoder(uhist, yhist)
        yest = model.h(zest)
        x_log[k + 1] = zest; y_log[k + 1] = y_meas
        y_est_log[k + 1] = model.ystd * yest[0] + model.y0
        u_log[k] = u_den; e_log[k] = eps
        x = np.concatenate([x[nx:], x[-nx:]])
        x0n = zest.copy()
        u = np.concatenate([u[nu:], u[-nu:]])
    return dict(x_log=x_log, u_log=u_log, e_log=e_log, lpv_counter=lpv_counter, y_log=y_log,
                y_est_log=y_est_log, Xs=Xs_all, Us=Us_all, plan_log=plan_log, abc_first=abc_first,
                y_ref_n=y_ref_n)
