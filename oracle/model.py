"""fp64 restatement of the learned state-space model: f(x,u), h(x), the encoder (observer)
net and their Jacobians.

Network structure (deepSI simple_res_net, as re-expressed by mpcUtil.py:6-82 of the
reference):   net(z) = W_lin z + b_lin + W4 tanh(W2 tanh(W0 z + b0) + b2) + b4
with z = [x; u] for f (mpcUtil.py:66), z = x for h (mpcUtil.py:41) and
z = [u_hist; y_hist] for the encoder (deepSI encoders.py forward, traceback in
Observer.ipynb cell 14).

Jacobian (what casadi.jacobian produces at NMPC_test.py:53-55):
    J(z) = W_lin + W4 diag(1-t2^2) W2 diag(1-t1^2) W0
The constant offsets f(0,0), h(0) ("correction", NMPC_test.py:38-43) do not change J.
"""
import os

import numpy as np

_NAMES = ("lin_w", "lin_b", "w0", "b0", "w2", "b2", "w4", "b4")
DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                               "nmpc_amd", "data", "disc_snr30.npz")


class ResNet:
    """Linear skip + 2-hidden-layer tanh MLP, evaluated in fp64 on arrays [..., n_in]."""

    def __init__(self, lin_w, lin_b, w0, b0, w2, b2, w4, b4):
        f = lambda a: np.asarray(a, dtype=np.float64)
        self.lin_w, self.lin_b = f(lin_w), f(lin_b)
        self.w0, self.b0, self.w2, self.b2, self.w4, self.b4 = f(w0), f(b0), f(w2), f(b2), f(w4), f(b4)
        self.n_in = self.w0.shape[1]
        self.n_out = self.w4.shape[0]
        self.hidden = self.w0.shape[0]

    def params(self):
        return [getattr(self, n) for n in _NAMES]

    def __call__(self, z):
        z = np.asarray(z, dtype=np.float64)
        t1 = np.tanh(z @ self.w0.T + self.b0)
        t2 = np.tanh(t1 @ self.w2.T + self.b2)
        return z @ self.lin_w.T + self.lin_b + t2 @ self.w4.T + self.b4

    def jac(self, z):
        """dnet/dz at z[..., n_in] -> [..., n_out, n_in]."""
        z = np.asarray(z, dtype=np.float64)
        t1 = np.tanh(z @ self.w0.T + self.b0)
        t2 = np.tanh(t1 @ self.w2.T + self.b2)
        d1 = 1.0 - t1 * t1
        d2 = 1.0 - t2 * t2
        # (W4 diag d2) (W2 diag d1) W0, batched over leading dims
        m2 = self.w4[None, :, :] * d2.reshape(-1, 1, self.hidden)          # [P, out, H]
        m1 = self.w2[None, :, :] * d1.reshape(-1, 1, self.hidden)          # [P, H, H]
        j = m2 @ m1 @ self.w0 + self.lin_w
        return j.reshape(z.shape[:-1] + (self.n_out, self.n_in))


class Model:
    """The encoder model: f, h, observer net, normalisation scalars."""

    def __init__(self, fn, hn, enc=None, u0=0.0, ustd=1.0, y0=0.0, ystd=1.0, na=4, nb=4):
        self.fn, self.hn, self.enc = fn, hn, enc
        self.nx = hn.n_in
        self.nu = fn.n_in - self.nx
        self.ny = hn.n_out
        self.u0, self.ustd, self.y0, self.ystd = float(u0), float(ustd), float(y0), float(ystd)
        self.na, self.nb = int(na), int(nb)
        zx, zu = np.zeros(self.nx), np.zeros(self.nu)
        self.f0 = self.f(zx, zu)          # "correction",   NMPC_test.py:38
        self.h0 = self.h(zx)              # "correction_h", NMPC_test.py:41

    @classmethod
    def load(cls, path=DEFAULT_WEIGHTS):
        d = np.load(path)
        nets = {}
        for net in ("fn", "hn", "enc"):
            nets[net] = ResNet(*[d["%s_%s" % (net, n)] for n in _NAMES])
        return cls(nets["fn"], nets["hn"], nets["enc"], d["u0"], d["ustd"], d["y0"], d["ystd"],
                   d["na"], d["nb"])

    @classmethod
    def synthetic(cls, nx, nu, ny, hidden, seed=0, n_hist=9):
        """Random nets with weights N(0, 1/fan_in); used for the wide-model configuration."""
        rng = np.random.default_rng(seed)

        def net(n_in, n_out):
            g = lambda r, c: (rng.standard_normal((r, c)) / np.sqrt(c)).astype(np.float32)
            b = lambda r: (0.1 * rng.standard_normal(r)).astype(np.float32)
            return ResNet(g(n_out, n_in), b(n_out), g(hidden, n_in), b(hidden),
                          g(hidden, hidden), b(hidden), g(n_out, hidden), b(n_out))
        return cls(net(nx + nu, nx), net(nx, ny), net(n_hist, nx))

    # --- model functions -------------------------------------------------------------
    def f(self, x, u):
        return self.fn(np.concatenate([np.asarray(x, float), np.asarray(u, float)], axis=-1))

    def h(self, x):
        return self.hn(x)

    def Jf(self, x, u):
        return self.fn.jac(np.concatenate([np.asarray(x, float), np.asarray(u, float)], axis=-1))

    def Jfx(self, x, u):
        return self.Jf(x, u)[..., :, :self.nx]

    def Jfu(self, x, u):
        return self.Jf(x, u)[..., :, self.nx:]

    def Jhx(self, x, u=None):
        return self.hn.jac(x)

    def encoder(self, uhist, yhist):
        """Observer: state estimate from past inputs and outputs (u first, then y)."""
        return self.enc(np.concatenate([np.asarray(uhist, float), np.asarray(yhist, float)], axis=-1))
