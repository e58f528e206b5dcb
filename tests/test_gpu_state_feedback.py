"""GPU parity tests of the state-feedback controllers (par_NMPC_linear, NMPC_linear: NMPC.py:374-650)
and the state-constrained target selector (getXsUs, mpcUtil.py:167-239) through the C ABI against the
fp64 oracle.  The reference's trained full-state model is not part of its repository, so the model is
the synthetic disc model of conftest.state_disc_model; bars as for the output-feedback loop:
targets 1e-5, closed-loop trajectories 1e-4."""
import numpy as np
import pytest
import torch

from oracle import lpv as olpv
from oracle.nmpc import state_nmpc_linear, target_selector_state
from oracle.plant import random_level_reference
from conftest import state_disc_model
from contextlib import nullcontext as _nullcontext

pytestmark = pytest.mark.gpu
XMIN, XMAX, UMIN, UMAX = [-8.0, -2.0], [8.0, 2.0], -6.0, 6.0


@pytest.fixture(scope="module")
def smodel():
    import nmpc_amd
    om, y0, ystd = state_disc_model()
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    enc = nmpc_amd.Encoder(params, nmpc_amd.Norm(om.u0, om.ustd, y0, ystd), device="cuda:0")
    return om, enc, y0, ystd


def test_getXsUs_vs_oracle(smodel):
    import nmpc_amd
    om, enc, y0, ystd = smodel
    B, Ns = 3, 45
    refs = np.stack([random_level_reference(Ns, [10, 15], [-1, 1], np.random.default_rng(40 + b)) for b in range(B)])
    refs[1] *= 2.5                                           # beyond the state box: the slack takes over
    yrn = (refs - y0[1]) / ystd[1]
    xmin = (np.array(XMIN) - y0) / ystd; xmax = (np.array(XMAX) - y0) / ystd
    umin, umax = (UMIN - om.u0) / om.ustd, (UMAX - om.u0) / om.ustd
    lam, w = nmpc_amd.quad_nodes("simpson", 20, 0.05)
    nodes, weights = olpv.merge_nodes(*olpv.quad_nodes("simpson", 20, 0.05))
    for max_lpv in (0, 10):        # 0: the schedule the reference's code executes; 10: iterated to convergence
        Xs, Us, Es, it = nmpc_amd.getXsUs(yrn, 2, 1, 1, Ns, umin, umax, xmin, xmax, nmpc_amd.GetABC(enc, lam, w), None,
                                          np.array([[0.0, 1.0]]), None, np.zeros(1), return_iters=True, max_lpv=max_lpv)
        for b in range(B):
            xo, uo, eo, ito = target_selector_state(om, yrn[b], Ns, umin, umax, xmin, xmax, np.array([[0.0, 1.0]]), nodes, weights,
                                                    as_executed=max_lpv == 0)
            assert np.abs(Xs[b].cpu().numpy() - xo).max() < 1e-5 and np.abs(Us[b].cpu().numpy() - uo).max() < 1e-5
            assert np.abs(Es[b].cpu().numpy() - eo).max() < 1e-5
            same = (it[b].cpu().numpy() == ito).mean()
            print("state target selector max_lpv=%d controller %d: pass counts equal in %.4f of the samples" % (max_lpv, b, same))
            assert same == 1.0 if max_lpv == 0 else same >= 0.97
    assert Es[1].max() > 1e-3 and float(Es[0].abs().max()) < 1e-9


def test_par_NMPC_linear_vs_oracle(smodel):
    """Tracking with soft state constraints, measurement noise on both states, up to 5 passes."""
    import nmpc_amd
    om, enc, y0, ystd = smodel
    B, Nsim, Nc = 3, 40, 10
    Q = np.diag([1.0, 1000.0])
    refs = np.stack([np.vstack([np.zeros(Nsim + Nc), random_level_reference(Nsim + Nc, [10, 15], [-1, 1], np.random.default_rng(60 + b))])
                     for b in range(B)])
    refs[:, 1] *= 0.6                                        # the synthetic model is the disc's linearisation at rest
    refs[2, 1] *= 1.3                                        # asks for angles outside the box below
    xmin, xmax = [-8.0, -0.5], [8.0, 0.5]
    noise = np.random.default_rng(5).standard_normal((Nsim, B, 2)) * np.array([0.05, 0.005])
    plant = nmpc_amd.NoisyUnbalancedDisc(dt=0.1, sigma_n=[0.05, 0.005])
    out = nmpc_amd.par_NMPC_linear(plant, enc, xmin, xmax, UMIN, UMAX, [0.0, 0.0], 0, Q, 1.0, 0.1, 0.05, 20, refs, Nc=Nc,
                                   Nsim=Nsim, max_iterations=5, noise=noise)
    x_log, u_log, e_log, comp_t, t, runtime, lpvc, refn, ct = out
    assert x_log.shape == (B, (Nsim + 1) * 2) and u_log.shape == (B, Nsim) and ct.shape == (4, Nsim * 5)
    for b in range(B):
        o = state_nmpc_linear(om, xmin, xmax, UMIN, UMAX, [0.0, 0.0], Q, 1.0, 0.1, 0.05, 20, refs[b], Nc=Nc, Nsim=Nsim,
                              max_iterations=5, noise=noise[:, b], y0=y0, ystd=ystd)
        assert np.abs(x_log[b].cpu().numpy().reshape(Nsim + 1, 2) - o["x_log"]).max() < 1e-4
        assert np.abs(u_log[b].cpu().numpy() - o["u_log"]).max() < 1e-4
        assert np.abs(e_log[b, 0].cpu().numpy() - o["e_log"]).max() < 1e-4
        same = lpvc[b].cpu().numpy() == o["lpv_counter"]
        print("controller %d: pass counts that differ from the oracle's: %.4f of %d" % (b, 1.0 - same.mean(), same.size))
        assert same.mean() >= 0.95               # stop tolerance 1e-5 (NMPC.py:609): at the fp32 embedding's rounding level
    assert float(e_log[2].abs().max()) > 1e-3               # the slack was needed for controller 2
    # un-batched call, as the reference makes it
    o1 = nmpc_amd.par_NMPC_linear(plant, enc, xmin, xmax, UMIN, UMAX, [0.0, 0.0], 0, Q, 1.0, 0.1, 0.05, 20, refs[0], Nc=Nc,
                                  Nsim=Nsim, max_iterations=5, noise=noise[:, :1])
    assert o1[0].shape == ((Nsim + 1) * 2,) and torch.equal(o1[1], u_log[0])


def test_NMPC_linear_vs_oracle(smodel):
    """Regulation to the origin with hard constraints (no slack variable): the input bound is active
    during the first steps."""
    import nmpc_amd
    om, enc, y0, ystd = smodel
    Nsim, Nc = 25, 10
    Q = np.diag([100.0, 100.0])
    x0 = [3.0, 1.2]
    plant = nmpc_amd.NoisyUnbalancedDisc(dt=0.1, sigma_n=[0.0, 0.0])
    out = nmpc_amd.NMPC_linear(plant, enc, XMIN, XMAX, -2.0, 2.0, x0, 0, Q, 1.0, 0.1, 0.05, 20, None, Nc=Nc, Nsim=Nsim,
                               max_iterations=5)
    o = state_nmpc_linear(om, XMIN, XMAX, -2.0, 2.0, x0, Q, 1.0, 0.1, 0.05, 20, None, Nc=Nc, Nsim=Nsim, max_iterations=5,
                          tol=1e-7, rho=None, y0=y0, ystd=ystd)
    assert np.abs(out[0].cpu().numpy().reshape(Nsim + 1, 2) - o["x_log"]).max() < 1e-4
    assert np.abs(out[1].cpu().numpy() - o["u_log"]).max() < 1e-4
    assert np.abs(out[1].cpu().numpy()).max() > 2.0 - 1e-6 and np.abs(out[1].cpu().numpy()).max() < 2.0 + 1e-9
    assert out[8].shape == (4,) and np.abs(o["x_log"][-1]).max() < 1e-3


def test_par_NMPC_linear_as_written(smodel):
    """NMPC.py:580 (input targets us[:Nc] at every step) and NMPC.py:591 (+ Psi' Us) to the letter: the device loop
    with as_written=True against the oracle with the same switch, and the switch matters (the default differs)."""
    import nmpc_amd
    om, enc, y0, ystd = smodel
    B, Nsim, Nc = 2, 40, 10
    Q = np.diag([1.0, 1000.0])
    refs = np.stack([np.vstack([np.zeros(Nsim + Nc), 0.6 * random_level_reference(Nsim + Nc, [10, 15], [-1, 1], np.random.default_rng(80 + b))])
                     for b in range(B)])
    refs[:, 1, :3] = 0.3                                     # a non-zero first target: us[:Nc] then is not zero
    xmin, xmax = [-8.0, -0.5], [8.0, 0.5]
    plant = nmpc_amd.NoisyUnbalancedDisc(dt=0.1, sigma_n=[0.0, 0.0])
    logs = {}
    for aw in (True, False):
        out = nmpc_amd.par_NMPC_linear(plant, enc, xmin, xmax, UMIN, UMAX, [0.0, 0.0], 0, Q, 1.0, 0.1, 0.05, 20, refs, Nc=Nc,
                                       Nsim=Nsim, max_iterations=5, as_written=aw, return_status=True)
        assert int(out[9].abs().max()) == 0
        logs[aw] = out[1].cpu().numpy()
        for b in range(B):
            o = state_nmpc_linear(om, xmin, xmax, UMIN, UMAX, [0.0, 0.0], Q, 1.0, 0.1, 0.05, 20, refs[b], Nc=Nc, Nsim=Nsim,
                                  max_iterations=5, y0=y0, ystd=ystd, as_written=aw)
            ex = np.abs(out[0][b].cpu().numpy().reshape(Nsim + 1, 2) - o["x_log"]).max()
            eu = np.abs(out[1][b].cpu().numpy() - o["u_log"]).max()
            same = (out[6][b].cpu().numpy() == o["lpv_counter"]).mean()
            print("as_written=%s controller %d: max |x - oracle| %.2e, |u - oracle| %.2e, pass counts equal %.3f" % (aw, b, ex, eu, same))
            assert ex < 1e-4 and eu < 1e-4 and same >= 0.95
    assert np.abs(logs[True] - logs[False]).max() > 1e-2


def test_failed_controller_does_not_disturb_the_batch(smodel):
    """NMPC_linear (hard constraints): one controller of 64 measures a state outside the hard state box in step 6, so
    its QP is infeasible there.  It is flagged (status 2) and holds its previous plan, shifted; the other 63 controllers'
    logs are bitwise those of the run in which nothing fails."""
    import nmpc_amd
    om, enc, y0, ystd = smodel
    B, Nsim, Nc, bad, kbad = 64, 14, 10, 37, 6
    Q = np.diag([100.0, 100.0])
    x0 = [3.0, 1.2]
    plant = nmpc_amd.NoisyUnbalancedDisc(dt=0.1, sigma_n=[0.0, 0.0])
    noise = np.zeros((Nsim, B, 2))
    runs = {}
    for fail in (False, True):
        nz = noise.copy()
        if fail:
            nz[kbad - 1, bad, 1] = 50.0              # the measurement taken in step kbad - 1 is x0 of step kbad
        with pytest.warns(RuntimeWarning) if fail else _nullcontext():
            out, ctrl = nmpc_amd.NMPC_linear(plant, enc, XMIN, XMAX, -2.0, 2.0, np.tile(x0, (B, 1)), 0, Q, 1.0, 0.1, 0.05, 20, None, Nc=Nc,
                                             Nsim=Nsim, max_iterations=5, noise=nz, return_status=True, return_controller=True)
        runs[fail] = (out, {k: v.clone() for k, v in ctrl.logs.items()})
    (o0, l0), (o1, l1) = runs[False], runs[True]
    st = o1[9].cpu().numpy()
    assert st.shape == (B, Nsim) and int(np.abs(o0[9].cpu().numpy()).max()) == 0
    assert st[bad, kbad] == 2 and (np.delete(st, bad, 0) == 0).all() and (st[bad, :kbad] == 0).all()
    others = [b for b in range(B) if b != bad]
    for k in ("x_log", "u_log", "e_log", "lpv_counter", "plan_log"):
        assert torch.equal(l0[k][others], l1[k][others]), k
    # the flagged step applied the second entry of the plan made in the step before
    plan = l1["plan_log"][bad].cpu().numpy()
    assert np.array_equal(plan[kbad, :-1], plan[kbad - 1, 1:]) and plan[kbad, -1] == plan[kbad - 1, -1]
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.NMPC_linear(plant, enc, XMIN, XMAX, -2.0, 2.0, np.tile(x0, (B, 1)), 0, Q, 1.0, 0.1, 0.05, 20, None, Nc=Nc, Nsim=Nsim,
                             max_iterations=5, noise=nz, on_qp_failure="raise")
