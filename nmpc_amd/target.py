"""Steady-state target selector with the reference's signature (mpcUtil.py:241-320 getXsUs_Cs),
batched over controllers on the GPU."""
import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, current_stream
from .mpcutil import _f64


class GetABC:
    """Handle standing for the reference's CasADi functions get_A / get_B / get_C
    (NMPC_test.py:60-62): the model plus the quadrature node table they were built with."""

    def __init__(self, model, lam, w):
        self.model = model
        self.lam = np.ascontiguousarray(lam, np.float32)
        self.w = np.ascontiguousarray(w, np.float32)


def _workspace(model, nb, max_lpv):
    """Scratch for one selector call from torch's caching allocator: no hipMalloc / hipFree / device
    synchronisation on the call path."""
    n = int(lib().nmpc_target_workspace_bytes(model.nx, int(nb), int(max_lpv)))
    return torch.empty(n, dtype=torch.uint8, device=model.device), n


def getXsUs_Cs(y_reference_list_normalize, nx, nu, ny, Nsim, u_min, u_max, y_min, y_max, get_A, get_B=None,
               get_C=None, f0=None, h0=None, *, max_lpv=0, tol=1e-4, return_iters=False):
    """y_reference_list_normalize [B, >= Nsim] (or 1-D).  get_A: a GetABC handle, or the model
    itself followed by (lam, w) in get_B / get_C.  f0 / h0 are accepted for signature
    compatibility; the model's own offsets f(0,0), h(0) are used.
    max_lpv = 0 (default): the pass schedule mpcUtil.py:288-313 executes - xs, us are views of the
    solution buffer that the next solve overwrites before `xold = np.copy(xs)` is taken, so the stop
    test passes from the second solve on: at most two passes for sample 0, one (embedded at the
    previous sample's target) for every later sample.  max_lpv >= 1: iterate every sample until
    (xs, us) move by less than tol, at most max_lpv passes (the reference's cap is 20).
    Returns (Xs [B, nx, Nsim], Us [B, nu, Nsim], Es [B, 1, Nsim]) as the reference lays them out."""
    if not isinstance(get_A, GetABC):
        get_A = GetABC(get_A, get_B, get_C)
    model = get_A.model
    assert (nx, nu, ny) == (model.nx, model.nu, model.ny)
    dev = model.device
    yr = _f64(y_reference_list_normalize, dev)
    single = yr.dim() == 1
    if single:
        yr = yr[None]
    yr = yr[:, :Nsim].contiguous()
    B = yr.shape[0]
    Xs = torch.empty(B, Nsim, nx, dtype=torch.float64, device=dev)
    Us = torch.empty(B, Nsim, dtype=torch.float64, device=dev)
    Es = torch.empty(B, Nsim, dtype=torch.float64, device=dev)
    iters = torch.zeros(B, Nsim, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    sc = lambda v: float(np.asarray(v, np.float64).reshape(-1)[0])
    with torch.cuda.device(dev):
        ws, nws = _workspace(model, B, max_lpv)
        rc = lib().nmpc_target_select_ws(model.handle, B, Nsim, ptr(yr), sc(u_min), sc(u_max), sc(y_min), sc(y_max),
                                         get_A.lam.ctypes.data_as(_lib.c_float_p), get_A.w.ctypes.data_as(_lib.c_float_p),
                                         get_A.lam.size, int(max_lpv), float(tol), ptr(Xs), ptr(Us), ptr(Es), ptr(iters),
                                         ptr(status), ptr(ws), nws, current_stream())
    check(rc, "nmpc_target_select_ws")
    if int(status.sum()):
        raise _lib.NmpcError("getXsUs_Cs: singular steady-state system for %d controllers" % int((status != 0).sum()))
    out = (Xs.transpose(1, 2).contiguous(), Us[:, None, :].contiguous(), Es[:, None, :].contiguous())
    if single:
        out = tuple(o[0] for o in out)
    if return_iters:
        return out + (iters[0] if single else iters,)
    return out


def getXsUs(y_reference_list_normalize, nx, nu, ny, Nsim, u_min, u_max, x_min, x_max, get_A, get_B=None, C=None,
            f0=None, h0=0.0, *, max_lpv=0, tol=1e-5, return_iters=False):
    """State-constrained target selector with the reference's signature (mpcUtil.py:167-239).
    get_A: a GetABC handle (or the model with (lam, w) in get_B).  C: output row(s) [ny, nx] (ny = 1),
    f0: equality offset (None: the model's f(0,0), what NMPC.py:542 passes), h0: output offset.
    max_lpv = 0 (default): the schedule the reference executes (its loop compares xs with a view of
    the buffer the next solve overwrites, mpcUtil.py:224-233: two passes at most for sample 0, one
    for every later sample); max_lpv >= 1: iterate until xs, us stop changing (the reference's cap is 10).
    Returns (Xs [B, nx, Nsim], Us [B, nu, Nsim], Es [B, 1, Nsim])."""
    if not isinstance(get_A, GetABC):
        get_A = GetABC(get_A, *get_B)
    model = get_A.model
    assert (nx, nu) == (model.nx, model.nu) and ny == 1
    dev = model.device
    yr = _f64(y_reference_list_normalize, dev)
    single = yr.dim() == 1
    if single:
        yr = yr[None]
    yr = yr[:, :Nsim].contiguous()
    B = yr.shape[0]
    Xs = torch.empty(B, Nsim, nx, dtype=torch.float64, device=dev)
    Us = torch.empty(B, Nsim, dtype=torch.float64, device=dev)
    Es = torch.empty(B, Nsim, dtype=torch.float64, device=dev)
    iters = torch.zeros(B, Nsim, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    vec = lambda v: np.ascontiguousarray(np.asarray(v, np.float64).reshape(-1))
    xmin, xmax, Cv = vec(x_min), vec(x_max), vec(C)
    f0v = None if f0 is None else vec(f0)
    dp = lambda a: None if a is None else a.ctypes.data_as(_lib.c_double_p)
    with torch.cuda.device(dev):
        ws, nws = _workspace(model, B, max_lpv)
        rc = lib().nmpc_target_select_state_ws(model.handle, B, Nsim, ptr(yr), float(u_min), float(u_max), dp(xmin),
                                               dp(xmax), dp(Cv), dp(f0v), float(np.asarray(h0).reshape(-1)[0]),
                                               get_A.lam.ctypes.data_as(_lib.c_float_p),
                                               get_A.w.ctypes.data_as(_lib.c_float_p), get_A.lam.size, int(max_lpv),
                                               float(tol), ptr(Xs), ptr(Us), ptr(Es), ptr(iters), ptr(status),
                                               ptr(ws), nws, current_stream())
    check(rc, "nmpc_target_select_state_ws")
    if int(status.sum()):
        raise _lib.NmpcError("getXsUs: singular steady-state system for %d controllers" % int((status != 0).sum()))
    out = (Xs.transpose(1, 2).contiguous(), Us[:, None, :].contiguous(), Es[:, None, :].contiguous())
    if single:
        out = tuple(o[0] for o in out)
    if return_iters:
        return out + (iters[0] if single else iters,)
    return out
