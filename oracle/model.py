"""fp64 restatement of the learned state-space model: f(x,u), h(x), the encoder (observer)
net and their Jacobians.

Network structure (deepSI simple_res_net, as re-expressed by mpcUtil.py:6-82 of the
reference):   net(z) = W_lin z + b_lin + W4 tanh(W2 tanh(W0 z + b0) + b2) + b4
(two hidden layers: CasADiFn / CasADiHn, mpcUtil.py:34,58; CasADiExp, mpcUtil.py:6-31, loops over the
model's f_n_hidden_layers, which ResNet follows for any count)
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
    """Linear skip + tanh MLP with L hidden layers (L = 2 for the trained model; mpcUtil.py:6-31 CasADiExp takes
    L from the model), evaluated in fp64 on arrays [..., n_in].  Arguments in the reference's parameter order:
    lin_w, lin_b, then (W, b) of each hidden layer, then (W, b) of the output layer."""

    def __init__(self, lin_w, lin_b, *rest):
        f = lambda a: np.asarray(a, dtype=np.float64)
        assert len(rest) >= 4 and len(rest) % 2 == 0
        self.lin_w, self.lin_b = f(lin_w), f(lin_b)
        self.hid = [[f(rest[i]), f(rest[i + 1])] for i in range(0, len(rest) - 2, 2)]
        self.out = [f(rest[-2]), f(rest[-1])]
        self.n_in = self.hid[0][0].shape[1]
        self.n_out = self.out[0].shape[0]
        self.hidden = self.hid[0][0].shape[0]
        self.n_hidden_layers = len(self.hid)

    # names of the two-hidden-layer case (net_non_lin.net.{0,2,4} of the model file)
    w0 = property(lambda s: s.hid[0][0], lambda s, v: s.hid[0].__setitem__(0, v))
    b0 = property(lambda s: s.hid[0][1], lambda s, v: s.hid[0].__setitem__(1, v))
    w2 = property(lambda s: s.hid[1][0], lambda s, v: s.hid[1].__setitem__(0, v))
    b2 = property(lambda s: s.hid[1][1], lambda s, v: s.hid[1].__setitem__(1, v))
    w4 = property(lambda s: s.out[0], lambda s, v: s.out.__setitem__(0, v))
    b4 = property(lambda s: s.out[1], lambda s, v: s.out.__setitem__(1, v))

    def params(self):
        return [self.lin_w, self.lin_b] + [a for wb in self.hid for a in wb] + list(self.out)

    def __call__(self, z):
        z = np.asarray(z, dtype=np.float64)
        t = z
        for w, b in self.hid:
            t = np.tanh(t @ w.T + b)
        return z @ self.lin_w.T + self.lin_b + t @ self.out[0].T + self.out[1]

    def jac(self, z):
        """dnet/dz at z[..., n_in] -> [..., n_out, n_in]:  W_lin + W_out D_L W_L ... D_1 W_0."""
        z = np.asarray(z, dtype=np.float64)
        zf = z.reshape(-1, self.n_in)
        t = zf
        m = None                                             # [P, H, n_in]: d t_l / d z
        for w, b in self.hid:
            t = np.tanh(t @ w.T + b)
            wm = w[None, :, :] if m is None else w[None, :, :] @ m
            m = (1.0 - t * t)[:, :, None] * wm
        j = self.out[0][None, :, :] @ m + self.lin_w
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
    def synthetic(cls, nx, nu, ny, hidden, seed=0, n_hist=9, f_layers=2, h_layers=2):
        """Random nets with weights N(0, 1/fan_in); used for the wide-model configuration.  f_layers / h_layers:
        hidden layers of the f and h nets (the observer net keeps two)."""
        rng = np.random.default_rng(seed)

        def net(n_in, n_out, layers=2):
            g = lambda r, c: (rng.standard_normal((r, c)) / np.sqrt(c)).astype(np.float32)
            b = lambda r: (0.1 * rng.standard_normal(r)).astype(np.float32)
            p = [g(n_out, n_in), b(n_out), g(hidden, n_in), b(hidden)]
            for _ in range(layers - 1):
                p += [g(hidden, hidden), b(hidden)]
            return ResNet(*(p + [g(n_out, hidden), b(n_out)]))
        return cls(net(nx + nu, nx, f_layers), net(nx, ny, h_layers), net(n_hist, nx))

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
