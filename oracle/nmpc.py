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
             (4,1)-(4,) broadcast at mpcUtil.py:254 followed by h[:,0] at :302 drops it;
           - the target selectors' pass schedule as their code EXECUTES it (getXsUs_Cs
             mpcUtil.py:302-310, getXsUs :224-233): xs/us are views of the solution buffer,
             which the next solve overwrites before the "old" values are taken, so the stop
             test passes from the second solve on - sample 0 takes at most two passes, every
             later sample one, embedded at the previous sample's target.  With this schedule
             the closed loop reproduces the reference's recorded Experiments/lpv_counter.npy
             in 450 of 450 steps (446 with every sample iterated to convergence).
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
                    max_lpv=20, tol=1e-4, lam=1000.0, as_executed=False):
    """Steady-state targets (xs, us) per reference sample, warm-started along the sequence.

    as_executed=False iterates every sample until (xs, us) stop changing (<= max_lpv passes), the
    evident intent of mpcUtil.py:288-313.  as_executed=True reproduces what that code DOES: xs and
    us are views of the buffer `xue` (mpcUtil.py:306-307) that the next `xue[:] = solve_qp(...)`
    (:302) overwrites BEFORE `xold = np.copy(xs)` (:304) is taken, so from the second solve on the
    stop test compares the new solution with itself and passes: sample 0 takes at most two passes
    and every later sample exactly one, embedded at the previous sample's target."""
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
            if as_executed and (j > 0 or i > 0):
                break
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
                       rule="simpson", dlam=None, targets=None, plant_x0=None, merge=True, target_as_executed=True,
                       embed_fn=None, timers=None, plant="disc"):
    """Closed loop of the output-constrained LPV-NMPC with observer, single controller.

    noise: optional array [Nsim] added to the measured angle (sigma_n of the plant).
    embed_fn: alternative to lpv.embed with the same signature (lpv.embed_functorch).
    plant: "disc" (the reference's plant with the observer in the loop) or "model": the model is its own plant,
    x+ = f(x, u), y = h(x) + noise, and the controller is handed the state (no observer) - the closed loop of
    models whose physical plant is not part of the hot path (noise is then in physical output units).
    timers: optional float array [Nsim, 4] that receives per-step seconds in the reference's buckets
    (NMPC_test.py:152-300): getAB (embedding), solve (condensing + QP + prediction), overhead, sim (plant).
    Returns a dict of logs named as the reference's return tuple (NMPC_test.py:320).
    """
    import time
    embed = lpv.embed if embed_fn is None else embed_fn
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
                                               nodes, weights, as_executed=target_as_executed)
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
        t_step = time.perf_counter(); t_ab = t_solve = 0.0
        Xs = Xs_all[:, k + 1:k + Nc + 1].T.reshape(nx * Nc)
        Us = Us_all[:, k:k + Nc].T.reshape(Nc * nu)
        while True:
            t0 = time.perf_counter()
            xsched = np.concatenate([x0n, x[:-nx]]).reshape(Nc, nx)
            A, B, C = embed(model, xsched, u.reshape(Nc, nu), nodes, weights, xc=x.reshape(Nc, nx))
            lA, lB, lC = A.reshape(Nc * nx, nx), B.reshape(Nc * nx, nu), C.reshape(Nc * ny, nx)
            t1 = time.perf_counter()
            if abc_first is None:
                abc_first = (lA.copy(), lB.copy(), lC.copy())
            u_old = u.copy()
            u, eps, x, _ = mpc_qp(lA, lB, lC, x0n, y_meas_n, Xs, Us, f0, h0, Psi, Omega, D, E, M, c,
                                  Nc, nx, nu, ny, rho)
            t_ab += t1 - t0; t_solve += time.perf_counter() - t1
            lpv_counter[k] += 1
            if lpv_counter[k] >= max_iterations or np.linalg.norm(u - u_old) < tol:
                break
        plan_log[k] = u
        u_den = model.ustd * u[0] + model.u0
        t0 = time.perf_counter()
        if plant == "model":
            zest = model.f(x0n, u[:nu])
            yest = model.h(zest)
            t_sim = time.perf_counter() - t0
            y_meas = model.ystd * yest[0] + model.y0 + (0.0 if noise is None else noise[k])
            y_meas_n = np.atleast_1d((y_meas - model.y0) / model.ystd)
        else:
            xp = disc_step(xp, u_den, dt)
            t_sim = time.perf_counter() - t0
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
        if timers is not None:
            tot = time.perf_counter() - t_step
            timers[k] = (t_ab, t_solve, tot - t_ab - t_solve - t_sim, t_sim)
    return dict(x_log=x_log, u_log=u_log, e_log=e_log, lpv_counter=lpv_counter, y_log=y_log,
                y_est_log=y_est_log, Xs=Xs_all, Us=Us_all, plan_log=plan_log, abc_first=abc_first,
                y_ref_n=y_ref_n)


# ------------------------------------------------------------------------------------------
# State-feedback variants (NMPC.py:374-650): the full state is measured, box constraints act on
# the predicted states through getDEMc, no observer and no output map.  The model offset is
# removed (rhs_c = rhs - f(0,0), NMPC.py:387,520), so predictions are Phi x0 + Gamma u.
#
# Departures from the code as written, all in favour of what the later output-feedback controller
# (NMPC_test.py) does for the same step:
#   - input targets follow the step, Us = us[k : k+Nc]  (NMPC.py:580 always takes the first Nc);
#   - linear cost term 2 (Gamma' Omega (Phi x0 - Xs) - Psi Us)  (NMPC.py:591 adds Psi' Us);
#     (both of these are restored by as_written=True: the fixed window us[:Nc] and + Psi' Us to the letter)
#   - the input warm start shifts by nu  (NMPC.py:495,640 shift it by nx);
#   - the last scheduling state after the shift is the predicted x_Nc  (NMPC.py:638 repeats x_{Nc-1});
def target_selector_state(model, y_ref_norm, Nsim, u_min, u_max, x_min, x_max, C, nodes, weights,
                          f0=None, h0=0.0, max_lpv=10, tol=1e-5, lam=100.0, as_executed=True):
    """mpcUtil.py:167-239 getXsUs: steady states with box constraints on (xs, us), fixed output
    matrix C; returns Xs [nx, Nsim], Us [nu, Nsim], Es [1, Nsim], iteration counts.
    as_executed: the pass schedule of the code as it runs (mpcUtil.py:224-233 compares xs with a view
    of the buffer the next solve overwrites: at most two passes for sample 0, one afterwards);
    False iterates every sample until xs, us stop changing."""
    nx, nu, ne = model.nx, model.nu, 1
    C = np.atleast_2d(np.asarray(C, float)); ny = C.shape[0]
    Q = np.eye(ny); R = np.eye(nu)
    Mi = np.vstack([np.zeros((nu, nx)), np.zeros((nu, nx)), -np.eye(nx), np.eye(nx)])
    Ei = np.vstack([-np.eye(nu), np.eye(nu), np.zeros((nx, nu)), np.zeros((nx, nu))])
    hvec = np.concatenate([[-u_min, u_max], -np.asarray(x_min, float), np.asarray(x_max, float)])
    f0 = np.asarray(model.f0 if f0 is None else f0, float).reshape(nx)
    h0 = np.atleast_1d(np.asarray(h0, float))
    nv = nx + nu + ne
    T = np.zeros((2 * (nx + nu), nv)); T[:, :nx] = Mi; T[:, nx:nx + nu] = Ei; T[:, nx + nu:] = -1.0
    P = np.zeros((nv, nv)); P[:nx, :nx] = C.T @ Q @ C; P[nx:nx + nu, nx:nx + nu] = R; P[nx + nu:, nx + nu:] = lam
    q = np.zeros(nv)
    Aeq = np.zeros((nx + ny, nv)); Aeq[nx:, :nx] = C
    beq = np.zeros(nx + ny); beq[:nx] = f0
    xs = np.zeros(nx); us = np.zeros(nu)
    Xs = np.zeros((nx, Nsim)); Us = np.zeros((nu, Nsim)); Es = np.zeros((ne, Nsim)); iters = np.zeros(Nsim, int)
    for j in range(Nsim):
        beq[nx:] = y_ref_norm[j] - h0
        q[:nx] = C.T @ Q @ (h0 - y_ref_norm[j])
        for i in range(max_lpv):
            As, Bs, _ = lpv.embed(model, xs, us, nodes, weights)
            Aeq[:nx, :nx] = np.eye(nx) - As
            Aeq[:nx, nx:nx + nu] = -Bs
            z = solve_qp(P, q, T, hvec, Aeq, beq)
            xold, uold = xs, us
            xs, us, e = z[:nx].copy(), z[nx:nx + nu].copy(), z[nx + nu:].copy()
            iters[j] = i + 1
            if as_executed and (j > 0 or i > 0):
                break
            if np.linalg.norm(xs - xold) <= tol and np.linalg.norm(us - uold) <= tol:
                break
        Xs[:, j], Us[:, j], Es[:, j] = xs, us, e
    return Xs, Us, Es, iters


def state_mpc_qp(list_A, list_B, x0, Xs, Us, Psi, Omega, D, E, M, c, Nc, nx, nu, rho, as_written=False):
    """Condensed QP of the state-feedback controllers (NMPC.py:446-458, 588-604).  rho = None: hard
    constraints, variables u (NMPC_linear); otherwise one slack with weight rho on every row.
    as_written: the linear term of NMPC.py:591, + Psi' Us."""
    Phi = mpcutil.getPhi(list_A, Nc, nx, nu)
    Gamma = mpcutil.getGamma(list_A, list_B, Nc, nx, nu)
    G = 2 * (Psi + Gamma.T @ Omega @ Gamma)
    F = 2 * (Gamma.T @ Omega @ (Phi @ x0 - Xs) + (Psi.T @ Us if as_written else -Psi @ Us))
    L = M @ Gamma + E
    hv = c[:, 0] - (D + M @ Phi) @ x0
    if rho is None:
        u = solve_qp(G, F, L, hv); eps = 0.0
        pieces = dict(Ge=G, Fe=F, Le=L, h=hv)
    else:
        nvar = Nc * nu + 1
        Ge = np.zeros((nvar, nvar)); Ge[:-1, :-1] = G; Ge[-1, -1] = rho
        Fe = np.append(F, 0.0); Le = np.hstack([L, -np.ones((L.shape[0], 1))])
        ue = solve_qp(Ge, Fe, Le, hv); u, eps = ue[:-1], ue[-1]
        pieces = dict(Ge=Ge, Fe=Fe, Le=Le, h=hv)
    return u, eps, Phi @ x0 + Gamma @ u, pieces


def state_nmpc_linear(model, x_min, x_max, u_min, u_max, x0, Q, R, dt, dlam, stages, x_reference_list=None,
                      Nc=5, Nsim=30, max_iterations=1, tol=1e-5, rho=10000.0, noise=None, y0=None, ystd=None,
                      C=(0.0, 1.0), targets=None, as_written=False):
    """Closed loop with measured state, single controller.
    x_reference_list [nx, >= Nsim+Nc] given: par_NMPC_linear (NMPC.py:507-650; tracking of the targets
    from getXsUs on row 1 of the reference, slack weight rho = 10000, stop 1e-5).
    x_reference_list None: NMPC_linear (NMPC.py:374-505; regulation to the origin, hard constraints:
    pass rho=None, stop 1e-7, at most 5 passes there).
    y0, ystd: state normalisation vectors (norm.y0, norm.ystd of a full-state model); noise [Nsim, nx].
    as_written: NMPC.py:580 (Us = us[:Nc] at every step) and NMPC.py:591 (+ Psi' Us) as the code has them.
    Returns a dict of logs."""
    nx, nu = model.nx, model.nu
    nodes, weights = lpv.merge_nodes(*lpv.quad_nodes("simpson", stages, dlam))   # lpv_int, NMPC.py:400,533
    y0 = np.zeros(nx) if y0 is None else np.asarray(y0, float).reshape(nx)
    ystd = np.ones(nx) if ystd is None else np.asarray(ystd, float).reshape(nx)
    x0 = np.asarray(x0, float)
    x0n = (x0 - y0) / ystd
    xmax_n = (np.asarray(x_max, float) - y0) / ystd; xmin_n = (np.asarray(x_min, float) - y0) / ystd
    umin_n = (u_min - model.u0) / model.ustd; umax_n = (u_max - model.u0) / model.ustd
    if x_reference_list is not None:
        refn = ((np.asarray(x_reference_list, float).T - y0) / ystd).T
        if targets is None:
            Xs_all, Us_all, _, _ = target_selector_state(model, refn[1], Nsim + Nc, umin_n, umax_n, xmin_n, xmax_n,
                                                         np.asarray(C, float), nodes, weights, h0=0.0)
        else:
            Xs_all, Us_all = targets
        u = np.tile((0 - model.u0) / model.ustd, Nc).astype(float)
    else:
        Xs_all = np.zeros((nx, Nsim + Nc + 1)); Us_all = np.zeros((nu, Nsim + Nc + 1))
        u = np.zeros(Nc * nu)
    Psi = mpcutil.getPsi(Nc, R); Omega = mpcutil.getOmega(Nc, Q)
    D, E, M, c = mpcutil.getDEMc(xmin_n, xmax_n, umin_n, umax_n, Nc, nx, nu)
    x = np.tile(x0n, Nc)                     # scheduling states x_0 .. x_{Nc-1}
    xp = x0.copy()
    u_log = np.zeros(Nsim); x_log = np.zeros((Nsim + 1, nx)); x_log[0] = x0; e_log = np.zeros(Nsim)
    lpv_counter = np.zeros(Nsim, int); plan_log = np.zeros((Nsim, Nc))
    for k in range(Nsim):
        Xs = Xs_all[:, k + 1:k + Nc + 1].T.reshape(nx * Nc)
        Us = (Us_all[:, :Nc] if as_written else Us_all[:, k:k + Nc]).T.reshape(Nc * nu)
        while True:
            A, B, _ = lpv.embed(model, x.reshape(Nc, nx), u.reshape(Nc, nu), nodes, weights)
            u_old = u.copy()
            u, eps, xpred, _ = state_mpc_qp(A.reshape(Nc * nx, nx), B.reshape(Nc * nx, nu), x[:nx], Xs, Us, Psi, Omega,
                                            D, E, M, c, Nc, nx, nu, rho, as_written=as_written)
            x[nx:] = xpred[:(Nc - 1) * nx]
            lpv_counter[k] += 1
            if lpv_counter[k] >= max_iterations or np.linalg.norm(u - u_old) < tol:
                break
        plan_log[k] = u
        u_den = model.ustd * u[0] + model.u0
        xp = disc_step(xp, u_den, dt)
        x_meas = xp + (0.0 if noise is None else noise[k])
        x0n = (x_meas - y0) / ystd
        x_log[k + 1] = x_meas; u_log[k] = u_den; e_log[k] = eps
        x = np.concatenate([x[nx:], xpred[-nx:]])        # x[nx:(Nc+1)nx] of the reference's longer buffer
        x[:nx] = x0n
        u = np.concatenate([u[nu:], u[-nu:]])
    return dict(x_log=x_log, u_log=u_log, e_log=e_log, lpv_counter=lpv_counter, Xs=Xs_all, Us=Us_all, plan_log=plan_log)
