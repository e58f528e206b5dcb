"""Exact fp64 solver for the small strictly convex QPs of the NMPC loop.

    min 1/2 z'Pz + q'z   s.t.  G z <= h,  A z = b

The reference hands these to qpsolvers (`qp.solve_qp(..., solver="osqp")` at
NMPC_test.py:233, NMPC.py:602,767, mpcUtil.py:224,302; `"quadprog"` at NMPC.py:458), an
un-vendored dependency.  The optimum of a strictly convex QP is unique, so the oracle targets
that optimum with a textbook Goldfarb-Idnani dual active-set iteration (the algorithm behind
quadprog), written for clarity: every iteration re-solves the working-set KKT system densely.
`kkt_residual` certifies a returned point independently of how it was found.
Equalities are eliminated first through a null-space basis, so P only needs to be positive
definite on the null space of A (the target selector's P is singular, mpcUtil.py:264-266,300).
"""
import numpy as np


class QPError(RuntimeError):
    pass


def _solve_ineq(P, q, G, h, tol, max_iter):
    n = P.shape[0]
    m = G.shape[0]
    L = np.linalg.cholesky(P)
    Pinv = np.linalg.inv(P)
    Pinv = 0.5 * (Pinv + Pinv.T)
    x = -np.linalg.solve(L.T, np.linalg.solve(L, q))
    W = []                      # working set (row indices of G)
    lam = np.zeros(0)           # multipliers of the working set
    N = -G.T                    # constraint normals, s_i = n_i'x + h_i >= 0
    gnorm = np.maximum(np.linalg.norm(G, axis=1), 1e-300)
    scale = 1.0 + np.abs(h)
    it = 0
    if m == 0:
        return x, np.zeros(0), 0
    while True:
        s = h - G @ x
        viol = s / gnorm
        viol[W] = 0.0
        p = int(np.argmin(viol))
        if s[p] >= -tol * scale[p]:
            out = np.zeros(m)
            out[W] = lam
            return x, out, it
        npl = N[:, p]
        lam_p = 0.0
        while True:
            it += 1
            if it > max_iter:
                raise QPError("dual active set: iteration limit")
            if W:
                Nw = N[:, W]
                S = Nw.T @ Pinv @ Nw
                r = np.linalg.solve(S, Nw.T @ (Pinv @ npl))
                z = Pinv @ (npl - Nw @ r)
            else:
                r = np.zeros(0)
                z = Pinv @ npl
            zn = float(z @ npl)
            # dual step bound
            t1, kdrop = np.inf, -1
            for k in range(len(W)):
                if r[k] > 1e-14 * (1 + abs(lam[k])) and lam[k] / r[k] < t1:
                    t1, kdrop = lam[k] / r[k], k
            primal = zn > 1e-13 * float(npl @ (Pinv @ npl))
            t2 = -(s[p]) / zn if primal else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                raise QPError("QP infeasible")
            if primal:
                x = x + t * z
            lam = lam - t * r
            lam_p += t
            if primal and t2 <= t1:
                W.append(p)
                lam = np.append(lam, lam_p)
                break
            W.pop(kdrop)
            lam = np.delete(lam, kdrop)
            s = h - G @ x


def solve_qp(P, q, G=None, h=None, A=None, b=None, solver=None, tol=1e-11, max_iter=None,
             return_dual=False):
    """Primal optimum (qpsolvers.solve_qp signature; `solver` accepted and ignored)."""
    P = np.asarray(P, np.float64); q = np.asarray(q, np.float64).reshape(-1)
    n = q.size
    G = np.zeros((0, n)) if G is None else np.asarray(G, np.float64).reshape(-1, n)
    h = np.zeros(0) if h is None else np.asarray(h, np.float64).reshape(-1)
    P = 0.5 * (P + P.T)
    if max_iter is None:
        max_iter = 20 * (n + G.shape[0]) + 50
    if A is not None and np.size(A):
        A = np.asarray(A, np.float64).reshape(-1, n); b = np.asarray(b, np.float64).reshape(-1)
        Qf, Rf = np.linalg.qr(A.T, mode="complete")
        r = A.shape[0]
        Y, Z = Qf[:, :r], Qf[:, r:]
        xp = Y @ np.linalg.solve(Rf[:r, :].T, b)
        Pr = Z.T @ P @ Z
        qr_ = Z.T @ (P @ xp + q)
        y, lam, it = _solve_ineq(Pr, qr_, G @ Z, h - G @ xp, tol, max_iter)
        x = xp + Z @ y
    else:
        x, lam, it = _solve_ineq(P, q, G, h, tol, max_iter)
    if return_dual:
        return x, lam, it
    return x


def kkt_residual(P, q, G, h, x, lam, A=None, b=None):
    """max of (stationarity on null(A), primal infeasibility, dual infeasibility,
    complementarity), each relative to the problem scale."""
    P = np.asarray(P, np.float64); q = np.asarray(q, np.float64).reshape(-1)
    G = np.asarray(G, np.float64).reshape(-1, q.size); h = np.asarray(h, np.float64).reshape(-1)
    g = P @ x + q + G.T @ lam
    if A is not None and np.size(A):
        A = np.asarray(A, np.float64).reshape(-1, q.size)
        Qf, _ = np.linalg.qr(A.T, mode="complete")
        g = Qf[:, A.shape[0]:].T @ g
        pe = np.max(np.abs(A @ x - np.asarray(b).reshape(-1)))
    else:
        pe = 0.0
    sc = 1.0 + np.max(np.abs(q)) + np.max(np.abs(P @ x))
    s = h - G @ x
    return max(np.max(np.abs(g)) / sc, pe, np.max(np.maximum(-s, 0) / (1 + np.abs(h)), initial=0.0),
               np.max(np.maximum(-lam, 0), initial=0.0) / sc, np.max(np.abs(lam * s), initial=0.0) / sc)
