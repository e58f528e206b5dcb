"""fp64 restatement of the reference's condensed prediction matrices and constraint
constants (mpcUtil.py:84-134, 322-362).

All three horizon matrices derive from ONE prefix-product scan of the per-step A_k
(list_A row-block k = A_k):     P_0 = I,   P_m = A_m P_{m-1}   (m = 1..Nc-1)
  getPhi   (mpcUtil.py:84-91)   block i      = A_i ... A_0            = P_i A_0
  getGamma (mpcUtil.py:93-104)  block (i,j)  = A_{i-j} ... A_1 B_j    = P_{i-j} B_j   (j <= i)
  getF0    (mpcUtil.py:322-331) block i      = (sum_{j<=i} A_j...A_1) f0 = (sum_{j<=i} P_j) f0
The Gamma/F0 index pattern is the reference's own (it is not the textbook LTV product) and is
reproduced as is.
"""
import numpy as np


def _blocks(list_M, Nc, rows):
    return np.asarray(list_M, np.float64).reshape(Nc, rows, -1)


def prefix_products(list_A, Nc, nx):
    A = _blocks(list_A, Nc, nx)
    P = np.empty((Nc, nx, nx))
    P[0] = np.eye(nx)
    for m in range(1, Nc):
        P[m] = A[m] @ P[m - 1]
    return P, A


def getPhi(list_A, Nc, nx, nu=None):
    P, A = prefix_products(list_A, Nc, nx)
    return (P @ A[0]).reshape(Nc * nx, nx)


def getGamma(list_A, list_B, Nc, nx, nu):
    P, _ = prefix_products(list_A, Nc, nx)
    Bk = _blocks(list_B, Nc, nx)
    Gamma = np.zeros((Nc * nx, Nc * nu))
    for i in range(Nc):
        for j in range(i + 1):
            Gamma[i * nx:(i + 1) * nx, j * nu:(j + 1) * nu] = P[i - j] @ Bk[j]
    return Gamma


def getF0(list_A, f0, Nc, nx):
    P, _ = prefix_products(list_A, Nc, nx)
    S = np.cumsum(P, axis=0).reshape(Nc * nx, nx)
    return S @ np.asarray(f0, np.float64).reshape(nx, -1)


def getZ(list_C, Nc, ny, nx):
    Ck = _blocks(list_C, Nc, ny)
    Z = np.zeros((Nc * ny, Nc * nx))
    for i in range(Nc):
        Z[i * ny:(i + 1) * ny, i * nx:(i + 1) * nx] = Ck[i]
    return Z


def getPsi(Nc, R):
    return np.kron(np.eye(Nc), np.atleast_2d(np.asarray(R, np.float64)))


def getOmega(Nc, Q):
    return np.kron(np.eye(Nc), np.atleast_2d(np.asarray(Q, np.float64)))


def _box(lo, hi, u_min, u_max, Nc, nz, nu):
    """D z0 + M Zstack + E U <= c for boxes lo <= z_i <= hi (i = 0..Nc), u_min <= u_i <= u_max.

    Row block i < Nc: [-u_i; u_i; -z_i; z_i]; z_0 enters through D, z_i (i >= 1) through
    column block i-1 of M; the terminal block [-z_Nc; z_Nc] uses column block Nc-1.
    Scalar input bounds (nu = 1), as in the reference (mpcUtil.py:113, 341).
    """
    lo = np.asarray(lo, np.float64).reshape(-1); hi = np.asarray(hi, np.float64).reshape(-1)
    assert nu == 1 and lo.size == nz and hi.size == nz
    rb = 2 * (nz + nu)
    nrow = Nc * rb + 2 * nz
    c = np.empty((nrow, 1)); D = np.zeros((nrow, nz)); M = np.zeros((nrow, Nc * nz)); E = np.zeros((nrow, Nc * nu))
    bi = np.concatenate([[-float(u_min), float(u_max)], -lo, hi])
    Iz = np.eye(nz)
    for i in range(Nc):
        r = i * rb
        c[r:r + rb, 0] = bi
        E[r, i] = -1.0; E[r + 1, i] = 1.0
        tgt, col = (D, 0) if i == 0 else (M, (i - 1) * nz)
        tgt[r + 2:r + 2 + nz, col:col + nz] = -Iz
        tgt[r + 2 + nz:r + rb, col:col + nz] = Iz
    r = Nc * rb
    c[r:, 0] = np.concatenate([-lo, hi])
    M[r:r + nz, (Nc - 1) * nz:] = -Iz
    M[r + nz:, (Nc - 1) * nz:] = Iz
    return D, E, M, c


def getDEMc(x_min, x_max, u_min, u_max, Nc, nx, nu):
    """State + input box (mpcUtil.py:112-134)."""
    return _box(x_min, x_max, u_min, u_max, Nc, nx, nu)


def getDEMc_out(y_min, y_max, u_min, u_max, Nc, ny, nu):
    """Output + input box (mpcUtil.py:340-362)."""
    return _box(y_min, y_max, u_min, u_max, Nc, ny, nu)
