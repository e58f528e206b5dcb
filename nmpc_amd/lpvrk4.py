"""lpv_rk4 with the reference's signature (lpv_rk4.py:1-11), batched, fp64 on the GPU."""
import torch

from ._lib import lib, check, ptr, current_stream
from .mpcutil import _f64


def lpv_rk4(x, u, A, B, dt, correction):
    """One classic RK4 step of xdot = A x + B u + correction with frozen A, B.
    x [..., nx], u [..., nu], A [..., nx, nx], B [..., nx, nu], correction [nx]."""
    A = _f64(A); dev = A.device
    nx = A.shape[-1]
    batch = A.shape[:-2]
    B = _f64(B, dev); nu = B.shape[-1]
    x = _f64(x, dev).reshape(batch + (nx,)); u = _f64(u, dev).reshape(batch + (nu,))
    c = _f64(correction, dev).reshape(nx)
    n = 1
    for s in batch:
        n *= int(s)
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = lib().nmpc_lpv_rk4(n, nx, nu, ptr(x), ptr(u), ptr(A), ptr(B), float(dt), ptr(c), ptr(out), current_stream())
    check(rc, "nmpc_lpv_rk4")
    return out
