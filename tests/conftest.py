import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return {n: np.load(os.path.join(GOLDEN, "ref_%s.npz" % n)) for n in ("condense", "lpv", "jac", "ablist")}


@pytest.fixture(scope="session")
def omodel():
    from oracle.model import Model
    return Model.load()


@pytest.fixture(scope="session")
def enc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nmpc_amd
    return nmpc_amd.load_system(device="cuda:0")


def disc_setup(omodel, Nc=10):
    """Normalised bounds / weights of the canonical configuration (NMPC_test.py:326-344)."""
    import numpy as np
    ymin = (np.array([-1.2]) - omodel.y0) / omodel.ystd
    ymax = (np.array([1.2]) - omodel.y0) / omodel.ystd
    umin = (-4 - omodel.u0) / omodel.ustd
    umax = (4 - omodel.u0) / omodel.ustd
    return dict(ymin=ymin, ymax=ymax, umin=umin, umax=umax, Q=np.diag([1000.0, 10.0]), R=1.0, rho=1000.0, Nc=Nc)


def state_disc_model(seed=0, nl_scale=0.05):
    """A full-state model of the unbalanced disc for the state-feedback controllers (the reference's
    trained state model, systems/UnbalancedDisk_dt01_e100_SNR_100, is not part of its repository):
    linear skip = the plant's discrete linearisation at rest in normalised coordinates, nonlinear
    branch = small random weights.  Returns (oracle Model, y0[2], ystd[2])."""
    import numpy as np
    from oracle.model import Model, ResNet
    from oracle.plant import disc_step
    y0 = np.zeros(2); ystd = np.array([5.0, 0.6]); u0, ustd = 0.0, 1.5
    eps = 1e-6; dt = 0.1
    f = lambda z: (disc_step(z[:2] * ystd + y0, z[2] * ustd + u0, dt) - y0) / ystd
    J = np.stack([(f(np.eye(3)[i] * eps) - f(-np.eye(3)[i] * eps)) / (2 * eps) for i in range(3)], axis=1)
    rng = np.random.default_rng(seed)
    g = lambda r, c: (rng.standard_normal((r, c)) / np.sqrt(c)).astype(np.float32)
    b = lambda r: (0.1 * rng.standard_normal(r)).astype(np.float32)
    fn = ResNet(J.astype(np.float32), np.zeros(2, np.float32), g(64, 3), b(64), g(64, 64), b(64), nl_scale * g(2, 64), b(2))
    hn = ResNet(np.array([[0.0, 1.0]], np.float32), np.zeros(1, np.float32), g(64, 2), b(64), g(64, 64), b(64),
                0.0 * g(1, 64), np.zeros(1, np.float32))
    return Model(fn, hn, None, u0=u0, ustd=ustd, y0=0.0, ystd=1.0), y0, ystd


def padded_disc_model(omodel, seed=5, extra=1):
    """The trained disc model embedded in a larger model (nx = 2 + extra, nu = ny = 1, H = 64): the first
    two states keep the trained nets, the extra ones are stable states coupled to them through small
    random weights in every net (f, h and the observer), so the closed loop behaves like the
    two-state one while every code path of the larger state dimension carries non-trivial numbers."""
    import numpy as np
    from oracle.model import Model, ResNet
    rng = np.random.default_rng(seed)
    f32 = lambda a: np.asarray(a, np.float32)
    sm = lambda *shape: f32(0.05 * rng.standard_normal(shape))
    fn, hn, en = omodel.fn, omodel.hn, omodel.enc
    e, nx = extra, 2 + extra
    # f: inputs (x1, x2, extra states, u), outputs nx
    lin_w = np.zeros((nx, nx + 1)); lin_w[:2, :2] = fn.lin_w[:, :2]; lin_w[:2, nx] = fn.lin_w[:, 2]
    lin_w[:2, 2:nx] = sm(2, e)
    lin_w[2:, :2] = 0.6 * sm(e, 2); lin_w[2:, nx] = 0.8 * sm(e)
    lin_w[2:, 2:nx] = np.diag(0.5 - 0.15 * np.arange(e)) + 0.5 * sm(e, e)
    w0 = np.concatenate([fn.w0[:, :2], sm(64, e), fn.w0[:, 2:3]], 1)
    w4 = np.concatenate([fn.w4, sm(e, 64) / 8], 0)
    f3 = ResNet(f32(lin_w), f32(np.concatenate([fn.lin_b, 0.01 * np.ones(e)])), f32(w0), f32(fn.b0), f32(fn.w2), f32(fn.b2),
                f32(w4), f32(np.concatenate([fn.b4, np.zeros(e)])))
    # h: inputs (x1, x2, extra states), output 1
    h3 = ResNet(f32(np.concatenate([hn.lin_w, sm(1, e)], 1)), f32(hn.lin_b), f32(np.concatenate([hn.w0, sm(64, e)], 1)),
                f32(hn.b0), f32(hn.w2), f32(hn.b2), f32(hn.w4), f32(hn.b4))
    # observer: histories -> nx states
    nh = en.w0.shape[1]
    e3 = ResNet(f32(np.concatenate([en.lin_w, 0.4 * sm(e, nh)], 0)), f32(np.concatenate([en.lin_b, np.zeros(e)])), f32(en.w0),
                f32(en.b0), f32(en.w2), f32(en.b2), f32(np.concatenate([en.w4, 0.4 * sm(e, 64)], 0)),
                f32(np.concatenate([en.b4, np.zeros(e)])))
    return Model(f3, h3, e3, u0=omodel.u0, ustd=omodel.ustd, y0=omodel.y0, ystd=omodel.ystd, na=omodel.na, nb=omodel.nb)


def narrowed_disc_model(omodel, hidden, alpha=0.5):
    """The trained disc model with `hidden` != 64 units per layer (the reference trains 32-wide encoders as well:
    UnbalancedDisc.ipynb, n_nodes_per_layer = 32).  hidden < 64: every net keeps its first `hidden` units with the
    output weights scaled by alpha.  hidden > 64: every net gains units with small random weights.  Either way the
    linear part is re-fitted so that value and Jacobian at the origin are those of the given net - the closed loop
    then behaves like the trained one near the origin while the nonlinear branch carries real numbers through every
    kernel.  Works on the padded models of padded_disc_model too."""
    import numpy as np
    from oracle.model import Model, ResNet
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    rng = np.random.default_rng(hidden)

    def resize(net):
        H0 = net.hidden
        if hidden <= H0:
            k = np.arange(hidden)
            w0, b0, w2, b2, w4 = net.w0[k], net.b0[k], net.w2[np.ix_(k, k)], net.b2[k], alpha * net.w4[:, k]
        else:
            e = hidden - H0
            w0 = np.vstack([net.w0, 0.3 * rng.standard_normal((e, net.n_in))])
            b0 = np.concatenate([net.b0, 0.1 * rng.standard_normal(e)])
            w2 = np.zeros((hidden, hidden)); w2[:H0, :H0] = net.w2
            w2[H0:, :] = rng.standard_normal((e, hidden)) / np.sqrt(hidden); w2[:H0, H0:] = 0.05 * rng.standard_normal((H0, e))
            b2 = np.concatenate([net.b2, 0.1 * rng.standard_normal(e)])
            w4 = np.hstack([net.w4, 0.05 * rng.standard_normal((net.n_out, e))])
        br = ResNet(np.zeros_like(net.lin_w), np.zeros_like(net.lin_b), f32(w0), f32(b0), f32(w2), f32(b2), f32(w4),
                    np.zeros_like(net.b4))
        z0 = np.zeros(net.n_in)
        return ResNet(f32(net.jac(z0) - br.jac(z0)), f32(net(z0) - br(z0)), br.w0, br.b0, br.w2, br.b2, br.w4, br.b4)
    return Model(resize(omodel.fn), resize(omodel.hn), resize(omodel.enc), omodel.u0, omodel.ustd, omodel.y0, omodel.ystd,
                 omodel.na, omodel.nb)
