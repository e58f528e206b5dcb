"""LPV embedding with the reference's call signatures (lpv_int.py, par_lpv_int.py, lpv_rk4.py and
mpcUtil.py:136-165), batched: x[..., nx], u[..., nu] are torch tensors on the GPU and every
leading dimension is a batch of independent scheduling points.

`Jfx`, `Jfu`, `Jhx` are model-bound handles (the reference passes CasADi Jacobian functions of
the same nets); the quadrature and the Jacobians are evaluated together by one HIP kernel, so
the handles only carry the model."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, current_stream


class Jacobian:
    """Handle standing for the reference's CasADi Function Jfx / Jfu / Jhx (NMPC_test.py:53-55)."""

    def __init__(self, model, kind):
        assert kind in ("fx", "fu", "hx")
        self.model, self.kind = model, kind


def jacobians(model):
    return Jacobian(model, "fx"), Jacobian(model, "fu"), Jacobian(model, "hx")


def quad_nodes(rule, stages, dlam=None):
    """Merged node table of a rule as fp32 host arrays (nmpc_quad_nodes)."""
    lam = np.zeros(256, np.float32); w = np.zeros(256, np.float32)
    code = {"simpson": 0, "trap": 1}[rule]
    n = lib().nmpc_quad_nodes(code, int(stages), float(dlam) if dlam else 0.0,
                              lam.ctypes.data_as(_lib.c_float_p), w.ctypes.data_as(_lib.c_float_p))
    if n < 0:
        raise _lib.NmpcError(lib().nmpc_last_error().decode())
    return lam[:n].copy(), w[:n].copy()


def _merge(nodes, weights):
    out = {}
    for n, w in zip(np.asarray(nodes, np.float64).tolist(), np.asarray(weights, np.float64).tolist()):
        out[n] = out.get(n, 0.0) + w
    return np.array(list(out.keys()), np.float32), np.array(list(out.values()), np.float32)


def _f32(t, model, last):
    t = torch.as_tensor(t, device=model.device)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.dim() == 0:
        t = t.reshape(1)
    assert t.shape[-1] == last, "last dimension must be %d, got %s" % (last, tuple(t.shape))
    return t.contiguous()


def embed(model, x=None, u=None, xc=None, lam=None, w=None, out=None):
    """A, B at (x, u) and/or C at xc for arbitrary leading batch dims; returns (A, B, C) with
    None for the parts not requested.  out: optional (A, B, C) fp32 device tensors to write into (entries for
    parts not requested are ignored), for callers that keep their buffers across calls."""
    lam = np.ascontiguousarray(lam, np.float32); w = np.ascontiguousarray(w, np.float32)
    A = B = C = None
    npts = None

    def _buf(i, shape):
        if out is None or out[i] is None:
            return torch.empty(shape, device=model.device, dtype=torch.float32)
        t = out[i]
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == int(np.prod(shape)) and t.device == model.device
        return t
    if x is not None:
        x = _f32(x, model, model.nx); u = _f32(u, model, model.nu)
        u = u.expand(x.shape[:-1] + (model.nu,)).contiguous() if u.shape[:-1] != x.shape[:-1] else u
        npts = x.numel() // model.nx
        A = _buf(0, x.shape[:-1] + (model.nx, model.nx))
        B = _buf(1, x.shape[:-1] + (model.nx, model.nu))
    if xc is not None:
        xc = _f32(xc, model, model.nx)
        nc = xc.numel() // model.nx
        assert npts is None or nc == npts, "x and xc must hold the same number of points"
        npts = nc
        C = _buf(2, xc.shape[:-1] + (model.ny, model.nx))
    with torch.cuda.device(model.device):
        rc = lib().nmpc_lpv_embed(model.handle, npts, ptr(x), ptr(u), ptr(xc),
                                  lam.ctypes.data_as(_lib.c_float_p), w.ctypes.data_as(_lib.c_float_p), lam.size,
                                  ptr(A), ptr(B), ptr(C), current_stream())
    check(rc, "nmpc_lpv_embed")
    return A, B, C


def lpv_int(x, nx, u, nu, Jfx, Jfu, dlam, stages):
    """lpv_int.py:4-32."""
    A, B, _ = embed(Jfx.model, x, u, None, *quad_nodes("simpson", stages, dlam))
    return A, B


def lpv_int_C(x, nx, u, nu, ny, Jfx, Jfu, Jhx, dlam, stages):
    """lpv_int.py:34-62 (C is evaluated at the same x)."""
    return embed(Jfx.model, x, u, x, *quad_nodes("simpson", stages, dlam))


def lambda_simpson(x, u, nx, nu, ny, Jfx, Jfu, Jhx, stages):
    """lpv_int.py:64-81."""
    return embed(Jfx.model, x, u, x, *quad_nodes("simpson", stages))


def lambda_trap(x, u, nx, nu, ny, Jfx, Jfu, Jhx, stages):
    """lpv_int.py:83-100."""
    return embed(Jfx.model, x, u, x, *quad_nodes("trap", stages))


def par_lpv_int(x, nx, u, nu, Jfx, Jfu, dlam, stages, L):
    """par_lpv_int.py:4-35: explicit node vector L = [k1 nodes | k2 nodes | k4 nodes]."""
    L = np.asarray(L, np.float64).reshape(-1)
    assert L.size == 3 * stages
    wts = np.concatenate([np.full(stages, 1.0), np.full(stages, 4.0), np.full(stages, 1.0)]) * (1 / 6 * dlam)
    A, B, _ = embed(Jfx.model, x, u, None, *_merge(L, wts))
    return A, B


class MappedGet:
    """Stands for the reference's horizon-mapped CasADi functions Get_A / Get_B / Get_C
    (`Function(...).map(Nc, "thread", 32)`, NMPC.py:82-86, NMPC_output.py:72-80): called with the
    scheduling states X [..., nx, Nc] (column k = x_k) and inputs u [..., Nc] it returns the Nc
    matrices side by side, [..., rows, Nc*cols], as the mapped function does."""

    def __init__(self, model, which, lam, w):
        assert which in ("A", "B", "C")
        self.model, self.which = model, which
        self.lam = np.ascontiguousarray(lam, np.float32)
        self.w = np.ascontiguousarray(w, np.float32)

    def __call__(self, X, u=None):
        m = self.model
        X = torch.as_tensor(X)
        pts = X.transpose(-1, -2)                                   # [..., Nc, nx]
        if self.which == "C":
            out = embed(m, None, None, pts, self.lam, self.w)[2]   # [..., Nc, ny, nx]
        else:
            uu = torch.as_tensor(u).reshape(pts.shape[:-1] + (m.nu,))
            out = embed(m, pts, uu, None, self.lam, self.w)[0 if self.which == "A" else 1]
        Nc = out.shape[-3]
        return out.transpose(-3, -2).reshape(out.shape[:-3] + (out.shape[-2], Nc * out.shape[-1]))


def mapped_get(model, rule="simpson", stages=20, dlam=None):
    """(Get_A, Get_B, Get_C) for a model and quadrature rule (lambda_simpson: NMPC_output.py:72)."""
    lam, w = quad_nodes(rule, stages, dlam)
    return tuple(MappedGet(model, k, lam, w) for k in "ABC")


def _split_states(x, Nc, nx):
    """np.vstack(np.split(x, Nc)).T of the reference: stacked [.., Nc*nx(,1)] -> [.., nx, Nc]"""
    x = torch.as_tensor(x)
    if x.shape[-1] == 1 and x.dim() >= 2 and x.shape[-2] == Nc * nx:
        x = x[..., 0]
    return x.reshape(x.shape[:-1] + (Nc, nx)).transpose(-1, -2)


def _fill(dst, p, Nc, rows, cols):
    """list[(rows*i):(rows*i+rows), :] = p[:, i*cols:(i+1)*cols] for every i (mpcUtil.py:138-139)"""
    blocks = p.reshape(p.shape[:-1] + (Nc, cols)).transpose(-3, -2)          # [..., Nc, rows, cols]
    src = blocks.reshape(p.shape[:-2] + (Nc * rows, cols)).to(dst.dtype)
    dst.copy_(src.to(dst.device))
    return dst


def getABlist(x, u, Nc, nx, nu, Get_A, Get_B, list_A, list_B):
    """mpcUtil.py:136-144: fills and returns the caller's list_A [..., Nc*nx, nx] and
    list_B [..., Nc*nx, nu] (torch tensors, written in place) from the stacked states
    x [..., Nc*nx(,1)] and inputs u [..., Nc]."""
    X = _split_states(x, Nc, nx)
    if isinstance(Get_A, MappedGet) and isinstance(Get_B, MappedGet) and Get_A.model is Get_B.model \
            and np.array_equal(Get_A.lam, Get_B.lam) and np.array_equal(Get_A.w, Get_B.w):
        m = Get_A.model                                              # one launch serves both
        pts = X.transpose(-1, -2)
        A, B, _ = embed(m, pts, torch.as_tensor(u).reshape(pts.shape[:-1] + (m.nu,)), None, Get_A.lam, Get_A.w)
        list_A.copy_(A.reshape(A.shape[:-3] + (Nc * nx, nx)).to(list_A.dtype))
        list_B.copy_(B.reshape(B.shape[:-3] + (Nc * nx, nu)).to(list_B.dtype))
        return list_A, list_B
    _fill(list_A, Get_A(X, u), Nc, nx, nx)
    _fill(list_B, Get_B(X, u), Nc, nx, nu)
    return list_A, list_B


def getABClist(x, u, Nc, nx, nu, ny, Get_A, Get_B, Get_C, list_A, list_B, list_C):
    """mpcUtil.py:147-158.  As in the reference, list_C is returned untouched (its fill is
    commented out there, mpcUtil.py:154-156; callers use getClist with the shifted states)."""
    getABlist(x, u, Nc, nx, nu, Get_A, Get_B, list_A, list_B)
    return list_A, list_B, list_C


def getClist(x, u, Nc, nx, ny, Get_C, list_C):
    """mpcUtil.py:160-165: list_C [..., Nc*ny, nx] from the stacked states x."""
    _fill(list_C, Get_C(_split_states(x, Nc, nx), u), Nc, ny, nx)
    return list_C
