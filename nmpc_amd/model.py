"""Host-side handle of the learned state-space model (the reference's `I_enc`, a deepSI
SS_encoder_general): attributes the controllers read (nx, nu, ny, na, nb, norm.{u0,ustd,y0,ystd})
plus the device-resident packed networks."""
import ctypes
import os

import numpy as np

from . import _lib

_NAMES = ("lin_w", "lin_b", "w0", "b0", "w2", "b2", "w4", "b4")
DISC_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "disc_snr30.npz")


class Norm:
    """deepSI System_data_norm: u_norm = (u - u0)/ustd, y_norm = (y - y0)/ystd."""

    def __init__(self, u0=0.0, ustd=1.0, y0=0.0, ystd=1.0):
        """y0 / ystd may be vectors (a full-state model's state normalisation, NMPC.py:524-525): they
        are kept as y0v / ystdv and the scalar fields hold their first entries."""
        self.u0, self.ustd = float(u0), float(ustd)
        y0a, ysa = np.atleast_1d(np.asarray(y0, np.float64)), np.atleast_1d(np.asarray(ystd, np.float64))
        self.y0, self.ystd = float(y0a[0]), float(ysa[0])
        if y0a.size > 1 or ysa.size > 1:
            self.y0v, self.ystdv = y0a.copy(), ysa.copy()


class Encoder:
    """fn / hn / encoder nets on the device.  `params` maps net name -> 4 + 2 L fp32 arrays in the
    reference's parameter order (mpcUtil.py:18-29): linear skip weight and bias, weight and bias of each of
    the L hidden layers, output weight and bias (L = 2 for the trained disc model; CasADiExp,
    mpcUtil.py:6-31, takes L from the model's f_n_hidden_layers)."""

    def __init__(self, params, norm=None, na=4, nb=4, device=None):
        import torch
        _lib.require_gpu()
        self.params = {k: [np.ascontiguousarray(a, dtype=np.float32) for a in v] for k, v in params.items()}
        fn, hn = self.params["fn"], self.params["hn"]
        self.hidden = fn[2].shape[0]
        self.nx = hn[2].shape[1]
        self.nu = fn[2].shape[1] - self.nx
        self.ny = hn[-2].shape[0]
        for name, lst in self.params.items():
            if len(lst) < 6 or len(lst) % 2:
                raise _lib.NmpcError("net %r: expected 4 + 2 L parameter arrays, got %d" % (name, len(lst)))
        self.f_n_hidden_layers = (len(fn) - 4) // 2
        self.h_n_hidden_layers = (len(hn) - 4) // 2
        self.na, self.nb = int(na), int(nb)
        self.norm = norm or Norm()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        enc = self.params.get("enc")
        self.n_enc_in = enc[2].shape[1] if enc is not None else 0

        def parr(lst):
            arr = (_lib.c_float_p * len(lst))()
            for i, a in enumerate(lst):
                arr[i] = a.ctypes.data_as(_lib.c_float_p)
            return arr
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            if enc is not None and len(enc) != 8:
                raise _lib.NmpcError("the observer net must have two hidden layers (8 parameter arrays)")
            rc = _lib.lib().nmpc_model_create_layers(ctypes.byref(h), self.nx, self.nu, self.ny, self.hidden,
                                                     self.f_n_hidden_layers, self.h_n_hidden_layers, self.n_enc_in,
                                                     parr(fn), parr(hn), parr(enc) if enc is not None else None)
        _lib.check(rc, "nmpc_model_create_layers")
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _lib.lib().nmpc_model_destroy(h)
            except Exception:
                pass
            self.handle = None

    @staticmethod
    def _net_at_zero(p):
        """The net at z = 0 in fp64 (host; a constant of the model): b_lin + W_out t_L + b_out with
        t_1 = tanh(b_0), t_l = tanh(W_l t_{l-1} + b_l)."""
        p = [np.asarray(a, np.float64) for a in p]
        t = np.tanh(p[3])
        for i in range(4, len(p) - 2, 2):
            t = np.tanh(p[i] @ t + p[i + 1])
        return p[1] + p[-2] @ t + p[-1]

    @property
    def f0(self):
        """f(0, 0): the reference's `correction` (NMPC_test.py:38), shape [nx]."""
        return self._net_at_zero(self.params["fn"])

    @property
    def h0(self):
        """h(0): the reference's `correction_h` (NMPC_test.py:41), shape [ny]."""
        return self._net_at_zero(self.params["hn"])

    @classmethod
    def from_npz(cls, path=DISC_WEIGHTS, device=None):
        d = np.load(path)
        params = {net: [d["%s_%s" % (net, n)] for n in _NAMES] for net in ("fn", "hn", "enc")
                  if "%s_w0" % net in d}
        norm = Norm(d["u0"], d["ustd"], d["y0"], d["ystd"]) if "u0" in d else Norm()
        return cls(params, norm, int(d["na"]) if "na" in d else 4, int(d["nb"]) if "nb" in d else 4, device)


def load_system(path=DISC_WEIGHTS, device=None):
    """Counterpart of deepSI.load_system (NMPC_test.py:362) for weight files in .npz form."""
    return Encoder.from_npz(path, device)
