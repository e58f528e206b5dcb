"""fp64 restatement of the reference's lambda-quadrature LPV embedding.

    A(p) = int_0^1 df/dx(lam x, lam u) dlam,   B likewise with df/du,   C with dh/dx

Every rule of the reference is a weighted sum of Jacobians at nodes lam_j p:
  * lpv_int / lpv_int_C   (lpv_int.py:4-62): composite Simpson, panels [lam_i, lam_i+dlam],
    lam_0 = 0, lam_{i+1} = lam_i + dlam (accumulated in floating point, lpv_int.py:30),
    independent `dlam` and `stages`;
  * lambda_simpson        (lpv_int.py:64-81): the same with dlam = 1/stages;
  * lambda_trap           (lpv_int.py:83-100): composite trapezoid, weights dlam/2;
  * par_lpv_int           (par_lpv_int.py:12-33): explicit node vector L laid out as
    [lam_i]_i, [lam_i+dlam/2]_i, [lam_i+dlam]_i, weights dlam/6 * (1, 4, 1).
`quad_nodes` returns the node/weight table as written (3*stages nodes for Simpson);
`merge_nodes` folds bit-identical nodes together (2*stages+1 unique ones).
"""
import numpy as np


def quad_nodes(rule, stages, dlam=None):
    """Nodes and weights exactly as the reference enumerates them."""
    if rule in ("simpson", "trap") and dlam is None:
        dlam = 1 / stages
    lam = 0
    nodes, weights = [], []
    for _ in range(int(stages)):
        if rule == "simpson":
            nodes += [lam, lam + dlam / 2, lam + dlam]
            weights += [dlam * 1 / 6, 4 * (dlam * 1 / 6), dlam * 1 / 6]
        elif rule == "trap":
            nodes += [lam, lam + dlam]
            weights += [dlam * 1 / 2, dlam * 1 / 2]
        else:
            raise ValueError(rule)
        lam = lam + dlam
    return np.array(nodes, dtype=np.float64), np.array(weights, dtype=np.float64)


def par_nodes(L, dlam, stages):
    """Node table of par_lpv_int: L = [k1 nodes | k2 nodes | k4 nodes]."""
    L = np.asarray(L, dtype=np.float64).reshape(-1)
    assert L.size == 3 * stages
    w = np.concatenate([np.full(stages, 1.0), np.full(stages, 4.0), np.full(stages, 1.0)]) * (1 / 6 * dlam)
    return L, w


def merge_nodes(nodes, weights):
    """Sum the weights of bit-identical nodes; keeps first-occurrence order."""
    out = {}
    for n, w in zip(nodes.tolist(), weights.tolist()):
        out[n] = out.get(n, 0.0) + w
    return np.array(list(out.keys())), np.array(list(out.values()))


def embed(model, x, u, nodes, weights, xc=None):
    """A, B at scheduling points (x,u) and C at points xc, for arrays of points.

    x [..., nx], u [..., nu], xc [..., nx] (defaults to x).  Returns A [..., nx, nx],
    B [..., nx, nu], C [..., ny, nx].  Summation runs over nodes in table order.
    """
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    xc = x if xc is None else np.asarray(xc, np.float64)
    nx = model.nx
    A = np.zeros(x.shape[:-1] + (nx, nx)); B = np.zeros(x.shape[:-1] + (nx, model.nu))
    C = np.zeros(xc.shape[:-1] + (model.ny, nx))
    for lam, w in zip(nodes, weights):
        J = model.Jf(lam * x, lam * u)
        A += w * J[..., :, :nx]
        B += w * J[..., :, nx:]
        C += w * model.Jhx(lam * xc)
    return A, B, C


def embed_functorch(model, x, u, nodes, weights, xc=None):
    """The same A, B, C with the Jacobians taken by torch.func.vmap(jacrev(.)) on the CPU, the
    mechanism of the reference's batched variant (NMPC_test.py:129-146, 176-177): one row per
    (scheduling point, node), fp64, then the weighted sums.  Used by the CPU baseline and to
    cross-check the analytic Jacobians."""
    import torch
    from torch.func import jacrev, vmap
    if not hasattr(model, "_torch_nets"):
        t = lambda a: torch.as_tensor(np.asarray(a, np.float64))

        def make(net):
            lw, lb, w0, b0, w2, b2, w4, b4 = (t(a) for a in net.params())
            return lambda z: lw @ z + lb + w4 @ torch.tanh(w2 @ torch.tanh(w0 @ z + b0) + b2) + b4
        fz, hz = make(model.fn), make(model.hn)
        model._torch_nets = (vmap(jacrev(lambda xx, uu: fz(torch.cat([xx, uu])), argnums=(0, 1))), vmap(jacrev(hz)))
    JacF, JacH = model._torch_nets
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    xc = x if xc is None else np.asarray(xc, np.float64)
    nx, nu, ny = model.nx, model.nu, model.ny
    lead = x.shape[:-1]
    X = x.reshape(-1, nx); U = u.reshape(-1, nu); XC = xc.reshape(-1, nx)
    lam = torch.as_tensor(np.asarray(nodes, np.float64)); wt = torch.as_tensor(np.asarray(weights, np.float64))
    P, S = X.shape[0], lam.numel()
    xt = (torch.as_tensor(X)[:, None, :] * lam[None, :, None]).reshape(P * S, nx)
    ut = (torch.as_tensor(U)[:, None, :] * lam[None, :, None]).reshape(P * S, nu)
    ct = (torch.as_tensor(XC)[:, None, :] * lam[None, :, None]).reshape(P * S, nx)
    fA, fB = JacF(xt, ut)
    fC = JacH(ct)
    A = (fA.reshape(P, S, nx, nx) * wt[None, :, None, None]).sum(1).numpy().reshape(lead + (nx, nx))
    B = (fB.reshape(P, S, nx, nu) * wt[None, :, None, None]).sum(1).numpy().reshape(lead + (nx, nu))
    C = (fC.reshape(P, S, ny, nx) * wt[None, :, None, None]).sum(1).numpy().reshape(xc.shape[:-1] + (ny, nx))
    return A, B, C


# --- reference-named entry points (numeric x, u; Jacobians as callables) ---------------
def _rule(x, u, Jfx, Jfu, Jhx, nodes, weights):
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    A = B = C = 0.0
    for lam, w in zip(nodes, weights):
        A = A + w * np.asarray(Jfx(lam * x, lam * u))
        B = B + w * np.asarray(Jfu(lam * x, lam * u))
        if Jhx is not None:
            C = C + w * np.asarray(Jhx(lam * x, lam * u))
    return A, B, C


def lpv_int(x, nx, u, nu, Jfx, Jfu, dlam, stages):
    A, B, _ = _rule(x, u, Jfx, Jfu, None, *quad_nodes("simpson", stages, dlam))
    return A, B


def lpv_int_C(x, nx, u, nu, ny, Jfx, Jfu, Jhx, dlam, stages):
    return _rule(x, u, Jfx, Jfu, Jhx, *quad_nodes("simpson", stages, dlam))


def lambda_simpson(x, u, nx, nu, ny, Jfx, Jfu, Jhx, stages):
    return _rule(x, u, Jfx, Jfu, Jhx, *quad_nodes("simpson", stages))


def lambda_trap(x, u, nx, nu, ny, Jfx, Jfu, Jhx, stages):
    return _rule(x, u, Jfx, Jfu, Jhx, *quad_nodes("trap", stages))


def par_lpv_int(x, nx, u, nu, Jfx, Jfu, dlam, stages, L):
    A, B, _ = _rule(x, u, Jfx, Jfu, None, *par_nodes(L, dlam, stages))
    return A, B


def lpv_rk4(x, u, A, B, dt, correction):
    """Classic RK4 step of xdot = A x + B u + c with frozen A, B (lpv_rk4.py:1-11); batched."""
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    c = np.asarray(correction, np.float64)
    bu = np.einsum("...ij,...j->...i", B, u) + c
    rhs = lambda s: dt * (np.einsum("...ij,...j->...i", A, s) + bu)
    k1 = rhs(x); k2 = rhs(x + k1 / 2); k3 = rhs(x + k2 / 2); k4 = rhs(x + k3)
    return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6


def get_ab_list(model, x, u, Nc, nodes, weights):
    """mpcUtil.py:136-144 getABlist: stacked states x[Nc*nx] (x_0 .. x_{Nc-1}) and inputs u[Nc] ->
    list_A [Nc*nx, nx], list_B [Nc*nx, nu] (row block k = A_k / B_k at (x_k, u_k))."""
    X = np.asarray(x, float).reshape(Nc, model.nx)
    A, B, _ = embed(model, X, np.asarray(u, float).reshape(Nc, model.nu), nodes, weights)
    return A.reshape(Nc * model.nx, model.nx), B.reshape(Nc * model.nx, model.nu)


def get_c_list(model, x, Nc, nodes, weights):
    """mpcUtil.py:160-165 getClist: list_C [Nc*ny, nx], row block k = C at the k-th stacked state."""
    X = np.asarray(x, float).reshape(Nc, model.nx)
    C = embed(model, X, np.zeros((Nc, model.nu)), nodes, weights)[2]
    return C.reshape(Nc * model.ny, model.nx)
