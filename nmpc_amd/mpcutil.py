"""Condensed prediction matrices and constraint constants with the reference's signatures
(mpcUtil.py:84-134, 322-362), batched over leading dimensions, fp64 on the GPU.

getPhi / getGamma / getF0 run the prefix-product kernel (nmpc_condense); getZ / getPsi /
getOmega / getDEMc / getDEMc_out only arrange constants and are plain tensor construction."""
import itertools

import torch

from ._lib import lib, check, ptr, current_stream, require_gpu


def _dev(device=None):
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _f64(t, device=None):
    t = torch.as_tensor(t)
    if not t.is_cuda:
        t = t.to(_dev(device))
    return t.to(torch.float64).contiguous()


def _condense(list_A, list_B, f0, Nc, nx, nu, want):
    lA = _f64(list_A)
    assert lA.shape[-2:] == (Nc * nx, nx), "list_A must be [..., Nc*nx, nx]"
    batch = lA.shape[:-2]
    nb = int(torch.tensor(batch).prod()) if len(batch) else 1
    lB = None
    if list_B is not None:
        lB = _f64(list_B, lA.device)
        assert lB.shape == batch + (Nc * nx, nu)
    f0t = _f64(f0, lA.device).reshape(-1) if f0 is not None else None
    Phi = torch.empty(batch + (Nc * nx, nx), dtype=torch.float64, device=lA.device) if "Phi" in want else None
    Gam = torch.empty(batch + (Nc * nx, Nc * nu), dtype=torch.float64, device=lA.device) if "Gamma" in want else None
    F0 = torch.empty(batch + (Nc * nx, 1), dtype=torch.float64, device=lA.device) if "F0" in want else None
    with torch.cuda.device(lA.device):
        rc = lib().nmpc_condense(nb, Nc, nx, nu, ptr(lA), ptr(lB), ptr(f0t), ptr(Phi), ptr(Gam), ptr(F0),
                                 current_stream())
    check(rc, "nmpc_condense")
    return Phi, Gam, F0


def getPhi(list_A, Nc, nx, nu):
    """Block i = A_i ... A_0 (mpcUtil.py:84-91)."""
    return _condense(list_A, None, None, Nc, nx, nu, ("Phi",))[0]


def getGamma(list_A, list_B, Nc, nx, nu):
    """Block (i,j) = A_{i-j} ... A_1 B_j for j <= i (mpcUtil.py:93-104)."""
    return _condense(list_A, list_B, None, Nc, nx, nu, ("Gamma",))[1]


def getF0(list_A, f0, Nc, nx):
    """Block i = (sum_{j<=i} A_j ... A_1) f0 (mpcUtil.py:322-331); returns [..., Nc*nx, 1]."""
    return _condense(list_A, None, f0, Nc, nx, 1, ("F0",))[2]


def getZ(list_C, Nc, ny, nx):
    """blkdiag(C_0 .. C_{Nc-1}) (mpcUtil.py:333-338)."""
    lC = _f64(list_C)
    batch = lC.shape[:-2]
    Ck = lC.reshape(batch + (Nc, ny, nx))
    Z = torch.zeros(batch + (Nc * ny, Nc * nx), dtype=torch.float64, device=lC.device)
    for i in range(Nc):
        Z[..., i * ny:(i + 1) * ny, i * nx:(i + 1) * nx] = Ck[..., i, :, :]
    return Z


def getPsi(Nc, R, device=None):
    """kron(I_Nc, R) (mpcUtil.py:106-107)."""
    R = torch.atleast_2d(torch.as_tensor(R, dtype=torch.float64))
    return torch.kron(torch.eye(Nc, dtype=torch.float64), R).to(_dev(device))


def getOmega(Nc, Q, device=None):
    """kron(I_Nc, Q) (mpcUtil.py:109-110)."""
    return getPsi(Nc, Q, device)


def _box(lo, hi, u_min, u_max, Nc, nz, nu, device):
    assert nu == 1, "scalar input bounds, as in the reference (mpcUtil.py:113)"
    lo = [float(v) for v in lo]; hi = [float(v) for v in hi]
    rb = 2 * (nz + nu)
    nrow = Nc * rb + 2 * nz
    bi = list(itertools.chain([-float(u_min), float(u_max)], [-v for v in lo], hi))
    c = torch.tensor(bi * Nc + [-v for v in lo] + hi, dtype=torch.float64).reshape(nrow, 1)
    D = torch.zeros(nrow, nz, dtype=torch.float64); M = torch.zeros(nrow, Nc * nz, dtype=torch.float64)
    E = torch.zeros(nrow, Nc * nu, dtype=torch.float64)
    Iz = torch.eye(nz, dtype=torch.float64)
    for i in range(Nc):
        r = i * rb
        E[r, i] = -1.0; E[r + 1, i] = 1.0
        tgt, col = (D, 0) if i == 0 else (M, (i - 1) * nz)
        tgt[r + 2:r + 2 + nz, col:col + nz] = -Iz
        tgt[r + 2 + nz:r + rb, col:col + nz] = Iz
    r = Nc * rb
    M[r:r + nz, (Nc - 1) * nz:] = -Iz
    M[r + nz:, (Nc - 1) * nz:] = Iz
    dev = _dev(device)
    return D.to(dev), E.to(dev), M.to(dev), c.to(dev)


def getDEMc(x_min, x_max, u_min, u_max, Nc, nx, nu, device=None):
    """State + input box constraints D x0 + M X + E U <= c (mpcUtil.py:112-134)."""
    return _box(x_min, x_max, u_min, u_max, Nc, nx, nu, device)


def getDEMc_out(y_min, y_max, u_min, u_max, Nc, ny, nu, device=None):
    """Output + input box constraints (mpcUtil.py:340-362)."""
    return _box(y_min, y_max, u_min, u_max, Nc, ny, nu, device)
