"""Degenerate sizes through the C ABI: empty batches, one controller, horizon 1, zero steps."""
import numpy as np
import pytest
import torch

from oracle.nmpc import output_nmpc_linear
from oracle.plant import random_level_reference

pytestmark = pytest.mark.gpu
Q = np.diag([1000.0, 10.0])


def test_empty_batches(enc):
    import nmpc_amd
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A, B, C = nmpc_amd.embed(enc, torch.zeros(0, 2), torch.zeros(0, 1), torch.zeros(0, 2), lam, w)
    assert A.shape == (0, 2, 2) and B.shape == (0, 2, 1) and C.shape == (0, 1, 2)
    assert nmpc_amd.getPhi(np.zeros((0, 20, 2)), 10, 2, 1).shape == (0, 20, 2)
    assert nmpc_amd.getGamma(np.zeros((0, 20, 2)), np.zeros((0, 20, 1)), 10, 2, 1).shape == (0, 20, 10)
    z = nmpc_amd.solve_qp(np.zeros((0, 3, 3)), np.zeros((0, 3)))
    assert z.shape == (0, 3)


@pytest.mark.parametrize("Nc", [1, 2])
def test_short_horizons_single_controller(enc, omodel, Nc):
    """B = 1, Nc = 1 and 2 (2- and 3-variable QPs, 6 and 10 rows) against the oracle."""
    import nmpc_amd
    Nsim = 12
    ref = random_level_reference(Nsim + Nc, [10, 15], [-1, 1], np.random.default_rng(8))
    o = output_nmpc_linear(omodel, -4, 4, [-1.2], [1.2], np.zeros(2), Q, 1.0, 0.1, 20, ref, Nc=Nc, Nsim=Nsim, max_iterations=4)
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[0.0])
    out, ctrl = nmpc_amd.output_NMPC_linear(plant, enc, -4, 4, [-1.2], [1.2], np.zeros(2), Q, 1.0, 0.1, 0.05, 20, ref, Nc=Nc,
                                            Nsim=Nsim, max_iterations=4, return_controller=True)
    assert np.abs(out[8].cpu().numpy() - o["y_log"]).max() < 1e-4 and np.abs(out[1].cpu().numpy() - o["u_log"]).max() < 1e-4
    before = {k: v.clone() for k, v in ctrl.logs.items()}
    ctrl.run(Nsim, 0)                                    # zero steps: nothing launched, nothing written
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], ctrl.logs[k]) for k in before)
    with pytest.raises(nmpc_amd.NmpcError):
        ctrl.run(Nsim, 1)                                # beyond n_sim


def test_run_in_pieces_equals_one_run(enc):
    """nmpc_ctrl_run over [0, Nsim) in one call and in uneven pieces (with the event profiling switched on
    for one of them) writes bitwise the same logs: the state carried between calls is complete."""
    import nmpc_amd
    B, Nc, Nsim = 700, 10, 24
    refs = np.stack([random_level_reference(Nsim + Nc, [10, 15], [-1, 1], np.random.default_rng(50 + b)) for b in range(B)])
    nm = enc.norm
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    ymin_n, ymax_n = (-1.2 - nm.y0) / nm.ystd, (1.2 - nm.y0) / nm.ystd
    umin_n, umax_n = (-4 - nm.u0) / nm.ustd, (4 - nm.u0) / nm.ustd
    yrn = torch.clamp((torch.as_tensor(refs, device="cuda") - nm.y0) / nm.ystd, ymin_n, ymax_n)
    Xs, Us, _ = nmpc_amd.getXsUs_Cs(yrn, 2, 1, 1, Nsim + Nc, umin_n, umax_n, [ymin_n], [ymax_n], nmpc_amd.GetABC(enc, lam, w))
    noise = torch.as_tensor(0.014 * np.random.default_rng(3).standard_normal((Nsim, B)), device="cuda")
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[0.014])

    def run(pieces):
        ctrl = nmpc_amd.Controller(enc, plant, B, Nc, Nsim, 4, Q, 1.0, umin_n, umax_n, ymin_n, ymax_n, lam, w, tol=1e-1, rho=1000.0)
        ctrl.set_targets(Xs.transpose(1, 2).contiguous(), Us[:, 0, :].contiguous(), noise)
        ctrl.reset((np.zeros(2) - nm.y0) / nm.ystd, (0 - nm.u0) / nm.ustd)
        k = 0
        for i, n in enumerate(pieces):
            ctrl.profile(1 if i == 1 else 0)
            ctrl.run(k, n)
            k += n
        torch.cuda.synchronize()
        ctrl.profile_read()
        return {name: t.clone() for name, t in ctrl.logs.items()}
    one, many = run([Nsim]), run([1, 7, 0, 11, 5])
    for name in one:
        assert torch.equal(one[name], many[name]), name


def test_controller_argument_errors(enc):
    import nmpc_amd
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[0.0])
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    mk = lambda lam_, w_: nmpc_amd.Controller(enc, plant, 4, 10, 5, 4, Q, 1.0, -2.0, 2.0, -2.0, 2.0, lam_, w_)
    with pytest.raises(nmpc_amd.NmpcError):
        mk(np.linspace(0, 1, 300), np.full(300, 1 / 300))          # more nodes than the library holds
    with pytest.raises(nmpc_amd.NmpcError):
        mk(lam, w[:-1])
    c = mk(lam, w)
    with pytest.raises(nmpc_amd.NmpcError):
        c.run(0, 1)                                                  # no targets yet


def test_horizons_beyond_the_lds_are_refused(enc):
    """The longest horizons: 60 at nx = 2 is the last whose condensing + QP tables fit a CU's LDS (tested against the
    oracle in test_gpu_closed_loop.py); 61..63 are refused with a message that says so, 64 (no lane left for the slack
    variable) by the range check - nothing is launched."""
    import nmpc_amd
    plant = nmpc_amd.OutputUnbalancedDisc(dt=0.1, sigma_n=[0.0])
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    nmpc_amd.Controller(enc, plant, 2, 60, 3, 2, Q, 1.0, -2.0, 2.0, -2.0, 2.0, lam, w)
    for Nc, word in ((61, "LDS"), (63, "LDS"), (64, "unsupported")):
        with pytest.raises(nmpc_amd.NmpcError, match=word):
            nmpc_amd.Controller(enc, plant, 2, Nc, 3, 2, Q, 1.0, -2.0, 2.0, -2.0, 2.0, lam, w)
