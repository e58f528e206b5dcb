"""solve_qp with the qpsolvers call signature the reference uses (NMPC_test.py:233,
NMPC.py:458,602,767, mpcUtil.py:224,302), batched over leading dimensions and solved on the GPU by
a dual active-set iteration to the exact optimum (fp64)."""
import torch

from ._lib import lib, check, ptr, current_stream, NmpcError
from .mpcutil import _f64

STATUS = {0: "ok", 1: "iteration limit", 2: "infeasible", 3: "P not positive definite"}


def solve_qp(P, q, G=None, h=None, A=None, b=None, solver=None, return_info=False, **_ignored):
    """min 1/2 z'Pz + q'z  s.t.  G z <= h, A z = b.  P [..., n, n], q [..., n] (or [..., n, 1]),
    G [..., m, n], h [..., m], A [..., meq, n], b [..., meq].  `solver` is accepted and ignored.
    Returns z [..., n]; with return_info also (multipliers [..., meq+m], status, iterations).
    Raises if any problem of the batch fails (the reference would crash on qpsolvers' None)."""
    P = _f64(P)
    n = P.shape[-1]
    batch = P.shape[:-2]
    nb = 1
    for s in batch:
        nb *= int(s)
    dev = P.device

    m = 0 if G is None else int(torch.as_tensor(G).shape[-2])
    meq = 0 if A is None else int(torch.as_tensor(A).shape[-2])
    G = None if G is None else _bcast(_f64(G, dev), batch)
    h = None if h is None else _bcast(_vec(_f64(h, dev), m), batch)
    A = None if A is None else _bcast(_f64(A, dev), batch)
    b = None if b is None else _bcast(_vec(_f64(b, dev), meq), batch)
    q = _bcast(_vec(_f64(q, dev), n), batch)
    z = torch.empty(batch + (n,), dtype=torch.float64, device=dev)
    lam = torch.empty(batch + (m + meq,), dtype=torch.float64, device=dev)
    status = torch.empty(batch, dtype=torch.int32, device=dev)
    iters = torch.empty(batch, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().nmpc_qp_solve(nb, n, m, meq, ptr(P), ptr(q), ptr(G), ptr(h), ptr(A), ptr(b), ptr(z), ptr(lam),
                                 ptr(status), ptr(iters), current_stream())
    check(rc, "nmpc_qp_solve")
    if return_info:
        return z, lam, status, iters
    bad = int((status != 0).sum())
    if bad:
        code = int(status[status != 0].reshape(-1)[0])
        raise NmpcError("solve_qp: %d of %d problems failed (%s)" % (bad, nb, STATUS.get(code, code)))
    return z


def _vec(t, n):
    if t.dim() >= 2 and t.shape[-1] == 1 and t.shape[-2] == n:
        t = t.reshape(t.shape[:-2] + (n,))
    return t


def _bcast(t, batch):
    core = t.dim() - len(batch)
    if core < 1 or tuple(t.shape[:len(batch)]) != tuple(batch):
        # same data for every problem of the batch
        kdim = t.dim() if t.dim() <= 2 else t.dim() - len(batch)
        t = t.reshape((1,) * len(batch) + tuple(t.shape[-kdim:])).expand(tuple(batch) + tuple(t.shape[-kdim:]))
    return t.contiguous()
