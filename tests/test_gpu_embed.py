"""GPU parity tests of the LPV embedding kernel through the C ABI (nmpc_lpv_embed) against the
fp64 oracle and the golden vectors of the reference's lpv_int.py.
Bar: A_k, B_k, C_k within 1e-5 relative (fp32 kernel)."""
import numpy as np
import pytest
import torch

from oracle import lpv as olpv

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(got, ref):
    """worst over points of ||got - ref||_max / ||ref||_max per matrix"""
    got = np.asarray(got, np.float64).reshape(ref.shape)
    num = np.abs(got - ref).reshape(ref.shape[0], -1).max(1)
    den = np.abs(ref).reshape(ref.shape[0], -1).max(1)
    return float((num / den).max())


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


@pytest.mark.parametrize("P", [1, 15, 16, 17, 33, 1000])
@pytest.mark.parametrize("rule,stages,dlam", [("simpson", 20, None), ("trap", 20, None), ("simpson", 7, 0.1)])
def test_embed_vs_oracle(enc, omodel, P, rule, stages, dlam):
    import nmpc_amd
    rng = np.random.default_rng(P * 31 + stages)
    x = f32(rng.uniform(-3, 3, (P, 2))); u = f32(rng.uniform(-3, 3, (P, 1))); xc = f32(rng.uniform(-3, 3, (P, 2)))
    lam, w = nmpc_amd.quad_nodes(rule, stages, dlam)
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    Ao, Bo, Co = olpv.embed(omodel, x, u, *olpv.merge_nodes(*olpv.quad_nodes(rule, stages, dlam)), xc=xc)
    assert relerr(A.cpu(), Ao) < TOL and relerr(B.cpu(), Bo) < TOL and relerr(C.cpu(), Co) < TOL


@pytest.mark.parametrize("n_nodes", [1, 3, 13, 64, 256])
def test_embed_arbitrary_node_tables(enc, omodel, n_nodes):
    """Node tables that are not equally spaced (random lambda in (0, 1], random weights, one node at 0 and
    one of zero weight mixed in): the kernel evaluates every exponential instead of stepping it, orders
    the nodes by weight and drops what contributes nothing; up to the maximum of 256 nodes."""
    import nmpc_amd
    rng = np.random.default_rng(n_nodes)
    lam = np.sort(rng.uniform(0.02, 1.0, n_nodes)).astype(np.float32); w = rng.uniform(0.1, 1.0, n_nodes).astype(np.float32)
    if n_nodes >= 13:
        lam[0] = 0.0; w[5] = 0.0
    w = (w / w.sum()).astype(np.float32)
    P = 40
    x = f32(rng.uniform(-3, 3, (P, 2))); u = f32(rng.uniform(-3, 3, (P, 1))); xc = f32(rng.uniform(-3, 3, (P, 2)))
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    Ao, Bo, Co = olpv.embed(omodel, x, u, lam.astype(np.float64), w.astype(np.float64), xc=xc)
    assert relerr(A.cpu(), Ao) < TOL and relerr(B.cpu(), Bo) < TOL and relerr(C.cpu(), Co) < TOL


def test_embed_far_out_of_range_points(enc, omodel):
    """Scheduling points thousands of standard deviations out (a diverged loop): pre-activations leave the
    range in which the kernel may step its exponentials from node to node, so those waves evaluate every
    node; the result still matches the oracle, and ordinary points in the same launch are unaffected."""
    import nmpc_amd
    rng = np.random.default_rng(5)
    P = 64
    x = f32(rng.uniform(-3, 3, (P, 2))); u = f32(rng.uniform(-3, 3, (P, 1))); xc = f32(rng.uniform(-3, 3, (P, 2)))
    x[16:32] *= 400.0; u[16:32] *= 400.0; xc[16:32] *= 400.0          # one tile far out
    x[40] *= 1500.0; xc[41] *= 1500.0                                   # single points inside ordinary tiles
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    Ao, Bo, Co = olpv.embed(omodel, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
    assert torch.isfinite(A).all() and torch.isfinite(B).all() and torch.isfinite(C).all()
    for got, ref in ((A, Ao), (B, Bo), (C, Co)):
        err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref)
        assert err.max() / np.abs(ref).max() < TOL


def test_reference_entry_points_on_golden(enc, golden):
    """lpv_int / lpv_int_C / lambda_simpson / lambda_trap / par_lpv_int with the reference's
    signatures reproduce what the reference's lpv_int.py returned for the same inputs."""
    import nmpc_amd
    l = golden["lpv"]
    x, u = torch.tensor(l["x"]), torch.tensor(l["u"])
    Jfx, Jfu, Jhx = nmpc_amd.jacobians(enc)
    A, B = nmpc_amd.lpv_int(x, 2, u, 1, Jfx, Jfu, 0.05, 20)
    assert relerr(A.cpu(), l["int_A"]) < TOL and relerr(B.cpu(), l["int_B"]) < TOL
    A, B, C = nmpc_amd.lpv_int_C(x, 2, u, 1, 1, Jfx, Jfu, Jhx, 0.05, 20)
    assert relerr(A.cpu(), l["intC_A"]) < TOL and relerr(C.cpu(), l["intC_C"]) < TOL
    A, B, C = nmpc_amd.lambda_simpson(x, u, 2, 1, 1, Jfx, Jfu, Jhx, 20)
    assert relerr(A.cpu(), l["simp_A"]) < TOL and relerr(B.cpu(), l["simp_B"]) < TOL and relerr(C.cpu(), l["simp_C"]) < TOL
    A, B, C = nmpc_amd.lambda_trap(x, u, 2, 1, 1, Jfx, Jfu, Jhx, 20)
    assert relerr(A.cpu(), l["trap_A"]) < TOL and relerr(B.cpu(), l["trap_B"]) < TOL and relerr(C.cpu(), l["trap_C"]) < TOL
    A, B = nmpc_amd.lpv_int(x, 2, u, 1, Jfx, Jfu, 0.1, 7)
    assert relerr(A.cpu(), l["odd_A"]) < TOL and relerr(B.cpu(), l["odd_B"]) < TOL
    L = np.concatenate([np.arange(20) * 0.05, np.arange(20) * 0.05 + 0.025, np.arange(20) * 0.05 + 0.05])
    A, B = nmpc_amd.par_lpv_int(x, 2, u, 1, Jfx, Jfu, 0.05, 20, L)
    assert relerr(A.cpu(), l["int_A"]) < TOL and relerr(B.cpu(), l["int_B"]) < TOL
    # single un-batched point, as the reference calls it
    A, B = nmpc_amd.lpv_int(x[3], 2, u[3], 1, Jfx, Jfu, 0.05, 20)
    assert A.shape == (2, 2) and B.shape == (2, 1)
    assert np.abs(A.cpu().numpy() - l["int_A"][3]).max() < 1e-5


def test_edge_cases(enc, omodel):
    import nmpc_amd
    # zero scheduling point: every node sees J(0); Observer.ipynb cell 12 (stages = 100 -> 201 nodes)
    lam, w = nmpc_amd.quad_nodes("simpson", 100)
    assert len(lam) == 201
    _, _, C = nmpc_amd.embed(enc, None, None, torch.zeros(1, 2), lam, w)
    assert np.abs(C.cpu().numpy() - [[-5.23925331, -2.59772804]]).max() < 1e-5
    # saturated tanh units, one-node rule, C-only and AB-only calls
    x = np.array([[50.0, -50.0], [1e-4, -1e-4], [0.0, 7.5]]); u = np.array([[30.0], [0.0], [-9.0]])
    for lam, w in ((np.array([1.0], np.float32), np.array([1.0], np.float32)), nmpc_amd.quad_nodes("simpson", 20)):
        A, B, C0 = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), None, lam, w)
        assert C0 is None
        Ao, Bo, Co = olpv.embed(omodel, f32(x), f32(u), lam.astype(float), w.astype(float))
        assert np.isfinite(A.cpu().numpy()).all()
        assert relerr(A.cpu(), Ao) < TOL and relerr(B.cpu(), Bo) < TOL
        _, _, C = nmpc_amd.embed(enc, None, None, torch.tensor(x), lam, w)
        assert relerr(C.cpu(), Co) < TOL
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.embed(enc, torch.zeros(2, 2), torch.zeros(2, 1), None, np.zeros(300, np.float32), np.zeros(300, np.float32))


def test_full_size_properties(enc, omodel):
    """B*Nc = 40960 points (the benchmark's launch): bitwise reproducible, independent of how the
    points are ordered into tiles, and correct on a sample."""
    import nmpc_amd
    P = 40960
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(P, 2, generator=g) * 4 - 2).cuda(); u = (torch.rand(P, 1, generator=g) * 6 - 3).cuda()
    xc = (torch.rand(P, 2, generator=g) * 4 - 2).cuda()
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A1, B1, C1 = nmpc_amd.embed(enc, x, u, xc, lam, w)
    A2, B2, C2 = nmpc_amd.embed(enc, x, u, xc, lam, w)
    assert torch.equal(A1, A2) and torch.equal(B1, B2) and torch.equal(C1, C2)
    perm = torch.randperm(P, generator=g).cuda()
    A3, B3, C3 = nmpc_amd.embed(enc, x[perm], u[perm], xc[perm], lam, w)
    assert torch.equal(A3, A1[perm]) and torch.equal(B3, B1[perm]) and torch.equal(C3, C1[perm])
    idx = torch.arange(0, P, 97)
    Ao, Bo, Co = olpv.embed(omodel, x[idx].cpu().double().numpy(), u[idx].cpu().double().numpy(),
                            *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc[idx].cpu().double().numpy())
    assert relerr(A1[idx].cpu(), Ao) < TOL and relerr(B1[idx].cpu(), Bo) < TOL and relerr(C1[idx].cpu(), Co) < TOL


@pytest.mark.parametrize("w2_scale", [1.0, 3e-3])
def test_synthetic_nets(w2_scale):
    """Random nets (not the trained model).  With W2 scaled down to ~1e-3 the fragments do not suit the
    split-fp16 format and the model is served by the fp32-MFMA kernel; the accuracy bar is the same."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(2, 1, 1, 64, seed=3)
    for net in (om.fn, om.hn):
        net.w2 = (net.w2 * w2_scale).astype(np.float32).astype(np.float64)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn), ("enc", om.enc))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    rng = np.random.default_rng(9)
    P = 200
    x = f32(rng.uniform(-2, 2, (P, 2))); u = f32(rng.uniform(-2, 2, (P, 1))); xc = f32(rng.uniform(-2, 2, (P, 2)))
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
    # In a random net W_lin and the nonlinear branch cancel at some points (|B| drops to 1e-2 of its
    # usual size), which inflates a per-matrix relative error without any loss in the kernel: measure
    # against the size the matrix has over the batch.
    for got, ref in ((A, Ao), (B, Bo), (C, Co)):
        err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
        assert err / np.abs(ref).max() < TOL
    assert relerr(A.cpu(), Ao) < TOL and relerr(C.cpu(), Co) < TOL


def test_model_offsets(enc, omodel):
    """Encoder.f0 / Encoder.h0: the reference's `correction` and `correction_h` (NMPC_test.py:38,41)."""
    assert np.abs(enc.f0 - omodel.f0).max() < 1e-12 and np.abs(enc.h0 - omodel.h0).max() < 1e-12
    assert enc.f0.shape == (2,) and enc.h0.shape == (1,)


def test_net_with_huge_bias():
    """A layer-1 bias of 60 (a unit pinned at +1 unless its input is enormous): such a model does not
    qualify for stepping the exponentials between nodes, and a point that un-pins the unit within
    the lambda-range is still integrated like the oracle does."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(2, 1, 1, 64, seed=4)
    om.fn.b0[3] = 60.0; om.hn.b0[7] = -60.0
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn), ("enc", om.enc))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    rng = np.random.default_rng(2)
    P = 48
    x = f32(rng.uniform(-2, 2, (P, 2))); u = f32(rng.uniform(-2, 2, (P, 1))); xc = f32(rng.uniform(-2, 2, (P, 2)))
    x[:16] *= 60.0; xc[:16] *= 60.0                     # lam * W0 z sweeps through -b0 for some of these
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
    for got, ref in ((A, Ao), (B, Bo), (C, Co)):
        err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
        assert err / np.abs(ref).max() < TOL


def test_ablist_family_on_golden(enc, golden):
    """getABlist / getABClist / getClist with the reference's signatures and in-place list buffers
    (mpcUtil.py:136-165), batched and un-batched, against what the reference returned."""
    import nmpc_amd
    g = golden["ablist"]
    Nc, nx, nu, ny = 10, 2, 1, 1
    Get_A, Get_B, Get_C = nmpc_amd.mapped_get(enc, "simpson", 20)
    x, u = torch.tensor(g["x"]), torch.tensor(g["u"])
    lA = torch.zeros(3, Nc * nx, nx, dtype=torch.float64, device="cuda"); lB = torch.zeros(3, Nc * nx, nu, dtype=torch.float64, device="cuda")
    lC = torch.full((3, Nc * ny, nx), 7.0, dtype=torch.float64, device="cuda")
    rA, rB, rC = nmpc_amd.getABClist(x, u, Nc, nx, nu, ny, Get_A, Get_B, Get_C, lA, lB, lC)
    assert rA is lA and rB is lB and rC is lC and bool((lC == 7.0).all())
    assert relerr(lA.cpu(), g["list_A"]) < TOL and relerr(lB.cpu(), g["list_B"]) < TOL
    nmpc_amd.getClist(x, u, Nc, nx, ny, Get_C, lC)
    assert relerr(lC.cpu(), g["list_C"]) < TOL
    # un-batched, separate (non-fused) Get_A / Get_B path, fp32 buffers
    a1 = torch.zeros(Nc * nx, nx, device="cuda"); b1 = torch.zeros(Nc * nx, nu, device="cuda")
    GA2 = nmpc_amd.MappedGet(enc, "A", *nmpc_amd.quad_nodes("simpson", 20))
    GB2 = nmpc_amd.MappedGet(enc, "B", *nmpc_amd.quad_nodes("trap", 20))
    nmpc_amd.getABlist(x[1], u[1], Nc, nx, nu, GA2, GB2, a1, b1)
    assert np.abs(a1.cpu().numpy() - g["list_A"][1]).max() < 1e-5
    pA = Get_A(x[1].reshape(Nc, nx).T, u[1])
    assert pA.shape == (nx, Nc * nx) and np.abs(pA[:, 2:4].cpu().numpy() - g["list_A"][1][2:4]).max() < 1e-5


@pytest.mark.parametrize("nx,nu,ny,H", [(4, 1, 1, 64), (8, 1, 3, 64), (3, 2, 2, 128), (8, 1, 1, 256), (5, 1, 2, 100)])
def test_wide_nets(nx, nu, ny, H):
    """Shapes beside the canonical one: nx = 4 at width 64 (the reference's LTI encoder, LTI.ipynb cell 5) on the
    split-fp16 tile as two tiles of two output rows; three outputs or two inputs (general-width exact-fp32 MFMA kernel
    with weight fragments streamed from L2); widths above 64 (the synthetic nx = 8 / H = 256 configuration) on the
    split-fp16 panel kernel.  H = 100 exercises the zero padding of hidden units."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(nx, nu, ny, H, seed=3)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    rng = np.random.default_rng(9)
    for P in (1, 37):
        x = f32(rng.uniform(-2, 2, (P, nx))); u = f32(rng.uniform(-2, 2, (P, nu))); xc = f32(rng.uniform(-2, 2, (P, nx)))
        lam, w = nmpc_amd.quad_nodes("simpson", 10)
        A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
        assert A.shape == (P, nx, nx) and B.shape == (P, nx, nu) and C.shape == (P, ny, nx)
        Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 10)), xc=xc)
        for got, ref in ((A, Ao), (B, Bo), (C, Co)):
            err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
            assert err / np.abs(ref).max() < TOL
    # reference-named entry point on a wide model
    Jfx, Jfu, Jhx = nmpc_amd.jacobians(enc)
    A1, B1, C1 = nmpc_amd.lambda_simpson(torch.tensor(x[0]), torch.tensor(u[0]), nx, nu, ny, Jfx, Jfu, Jhx, 10)
    assert np.abs(A1.cpu().numpy() - Ao[0]).max() / np.abs(Ao).max() < TOL
    # the synthetic model has no observer net: the closed loop refuses it
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.Controller(enc, nmpc_amd.OutputUnbalancedDisc(dt=0.1), 4, 10, 5, 4, np.eye(nx), 1.0, -1, 1, -1, 1, lam, w)


@pytest.mark.parametrize("nx,nu,ny,H,Lf,Lh", [(2, 1, 1, 64, 1, 1), (2, 1, 1, 64, 3, 3), (4, 1, 1, 64, 4, 4), (3, 2, 2, 128, 1, 3),
                                               (8, 1, 3, 128, 3, 1), (5, 1, 2, 100, 3, 2), (2, 1, 1, 256, 1, 2)])
def test_hidden_layer_counts(nx, nu, ny, H, Lf, Lh):
    """Nets with 1, 3 or 4 hidden layers, as CasADiExp (mpcUtil.py:6-31, used by par_NMPC_linear at NMPC.py:517)
    accepts them through f_n_hidden_layers: parameters in its order, embedding against the fp64 oracle at the same
    bar as the two-layer nets, f(0,0) / h(0) against the oracle."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(nx, nu, ny, H, seed=7, f_layers=Lf, h_layers=Lh)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    assert len(params["fn"]) == 4 + 2 * Lf and len(params["hn"]) == 4 + 2 * Lh
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    assert enc.f_n_hidden_layers == Lf and enc.h_n_hidden_layers == Lh
    assert np.abs(enc.f0 - om.f0).max() < 1e-12 and np.abs(enc.h0 - om.h0).max() < 1e-12
    rng = np.random.default_rng(9)
    for P, stages in ((1, 10), (53, 20)):
        x = f32(rng.uniform(-2, 2, (P, nx))); u = f32(rng.uniform(-2, 2, (P, nu))); xc = f32(rng.uniform(-2, 2, (P, nx)))
        lam, w = nmpc_amd.quad_nodes("simpson", stages)
        A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
        Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", stages)), xc=xc)
        for got, ref in ((A, Ao), (B, Bo), (C, Co)):
            err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
            assert err / np.abs(ref).max() < TOL
    A2, B2, C2 = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    assert torch.equal(A, A2) and torch.equal(B, B2) and torch.equal(C, C2)


def test_hidden_layer_limits():
    """More than two hidden layers of more than 128 units do not fit the kernel's LDS plan: refused at creation."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(2, 1, 1, 256, seed=1, f_layers=3, h_layers=2)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.Encoder(params, device="cuda:0")
    om = Model.synthetic(2, 1, 1, 32, seed=1, f_layers=5, h_layers=2)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    with pytest.raises(nmpc_amd.NmpcError):
        nmpc_amd.Encoder(params, device="cuda:0")


def test_wide_kernel_on_the_disc_model(enc, omodel, monkeypatch):
    """The general-width kernel is a drop-in for the 64-wide one: same model, same bar."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); import numpy as np, torch, nmpc_amd\n"
            "from oracle.model import Model; from oracle import lpv as olpv\n"
            "enc = nmpc_amd.load_system(device='cuda:0'); om = Model.load(); rng = np.random.default_rng(1)\n"
            "x = rng.uniform(-3, 3, (50, 2)).astype(np.float32).astype(float); u = rng.uniform(-3, 3, (50, 1)).astype(np.float32).astype(float)\n"
            "lam, w = nmpc_amd.quad_nodes('simpson', 20)\n"
            "A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(x), lam, w)\n"
            "Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes('simpson', 20)))\n"
            "e = max(np.abs(g.cpu().numpy().reshape(r.shape) - r).max() / np.abs(r).max() for g, r in ((A, Ao), (B, Bo), (C, Co)))\n"
            "print('ERR', e); sys.exit(0 if e < 1e-5 else 3)\n")
    env = dict(os.environ, NMPC_EMBED="wide")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("nx,nu,ny,H", [(8, 1, 1, 256), (3, 2, 2, 128), (5, 1, 2, 100), (8, 1, 3, 256)])
def test_wide_kernel_variants(nx, nu, ny, H, tmp_path):
    """Nets of hidden width 65..256 run on the split-fp16 panel kernel (three fp16 MFMA products per fp32 product,
    activations shared through LDS) unless NMPC_EMBED_WIDE=f32 keeps them on the exact-fp32 streaming kernel: both
    within 2e-6 of the fp64 oracle and of each other on 1000 points (ragged against the 16-, 32- and 64-point
    workgroups), relative to the size each matrix has over the batch; 1e-5 is the bar (lpv_int.py:64-81)."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, '.'); import numpy as np, torch, nmpc_amd\n"
            "from oracle.model import Model\n"
            "nx, nu, ny, H = %d, %d, %d, %d\n"
            "om = Model.synthetic(nx, nu, ny, H, seed=5)\n"
            "params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (('fn', om.fn), ('hn', om.hn))}\n"
            "enc = nmpc_amd.Encoder(params, device='cuda:0'); rng = np.random.default_rng(21); P = 1000\n"
            "x = rng.uniform(-2, 2, (P, nx)).astype(np.float32); u = rng.uniform(-2, 2, (P, nu)).astype(np.float32)\n"
            "xc = rng.uniform(-2, 2, (P, nx)).astype(np.float32)\n"
            "lam, w = nmpc_amd.quad_nodes('simpson', 10)\n"
            "A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)\n"
            "np.savez(sys.argv[1], x=x, u=u, xc=xc, A=A.cpu().numpy(), B=B.cpu().numpy(), C=C.cpu().numpy())\n" % (nx, nu, ny, H))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, envadd in (("panel", {}), ("f32", {"NMPC_EMBED_WIDE": "f32"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("NMPC_EMBED")}
        env.update(envadd)
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out[name] = np.load(f)
    from oracle.model import Model
    om = Model.synthetic(nx, nu, ny, H, seed=5)
    d = out["panel"]
    Ao, Bo, Co = olpv.embed(om, d["x"].astype(np.float64), d["u"].astype(np.float64),
                            *olpv.merge_nodes(*olpv.quad_nodes("simpson", 10)), xc=d["xc"].astype(np.float64))
    for key, ref in (("A", Ao), ("B", Bo), ("C", Co)):
        scale = np.abs(ref).max()
        for name in out:
            err = np.abs(out[name][key].astype(np.float64).reshape(ref.shape) - ref).max() / scale
            assert err < 2e-6, (name, key, err)
        assert np.abs(out["panel"][key].astype(np.float64) - out["f32"][key]).max() / scale < 2e-6
    assert not np.array_equal(out["panel"]["A"], out["f32"]["A"])      # two different kernels did run


def test_panel_kernel_ragged_batches():
    """The split-fp16 panel kernel works in workgroups of 16, 32 or 64 points (by output count): batches that end inside
    a workgroup, a single point, an empty batch and node counts that do not fill a node group give, point for point,
    bitwise the results of the full batch - a point's result does not depend on what shares its workgroup."""
    import nmpc_amd
    from oracle.model import Model
    for nx, nu, ny, H in ((8, 1, 1, 256), (3, 2, 2, 128)):
        om = Model.synthetic(nx, nu, ny, H, seed=11)
        params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
        enc = nmpc_amd.Encoder(params, device="cuda:0")
        rng = np.random.default_rng(4)
        P = 131
        x = torch.tensor(f32(rng.uniform(-2, 2, (P, nx)))); u = torch.tensor(f32(rng.uniform(-2, 2, (P, nu))))
        xc = torch.tensor(f32(rng.uniform(-2, 2, (P, nx))))
        for stages in (10, 3, 1):                       # 20, 6 and 2 nodes: whole groups, a partial group, less than a group
            lam, w = nmpc_amd.quad_nodes("simpson", stages)
            A, B, C = nmpc_amd.embed(enc, x, u, xc, lam, w)
            for n in (0, 1, 15, 16, 17, 63, 64, 65, 130):
                An, Bn, Cn = nmpc_amd.embed(enc, x[:n], u[:n], xc[:n], lam, w)
                assert An.shape == (n, nx, nx) and Bn.shape == (n, nx, nu) and Cn.shape == (n, ny, nx)
                assert torch.equal(An, A[:n]) and torch.equal(Bn, B[:n]) and torch.equal(Cn, C[:n])
            Ao, Bo, Co = olpv.embed(om, x.numpy().astype(np.float64), u.numpy().astype(np.float64),
                                    *olpv.merge_nodes(*olpv.quad_nodes("simpson", stages)), xc=xc.numpy().astype(np.float64))
            for got, ref in ((A, Ao), (B, Bo), (C, Co)):
                assert np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max() / np.abs(ref).max() < 2e-6


def test_embedding_kernel_variants_on_the_trained_model(omodel, tmp_path):
    """The three kernels that can serve the disc model - split-fp16 MFMA (default, with and without the stepped
    layer-1 exponentials), fp32 MFMA (NMPC_EMBED=f32), general-width (NMPC_EMBED=wide) - on the trained weights and
    the same 40 960 seeded scheduling points (the size of one full pass of the benchmark): each within 1e-6 of the
    fp64 oracle (lpv_int.py:64-81 lambda_simpson, stages = 20) on every eighth point, and within 2e-6 of each other
    on all of them, measured per matrix against the size the matrix has over the batch."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, '.'); import numpy as np, torch, nmpc_amd\n"
            "enc = nmpc_amd.load_system(device='cuda:0'); rng = np.random.default_rng(77); P = 40960\n"
            "x = rng.uniform(-2.5, 2.5, (P, 2)).astype(np.float32); u = rng.uniform(-2.5, 2.5, (P, 1)).astype(np.float32)\n"
            "xc = rng.uniform(-2.5, 2.5, (P, 2)).astype(np.float32)\n"
            "lam, w = nmpc_amd.quad_nodes('simpson', 20)\n"
            "A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)\n"
            "np.savez(sys.argv[1], x=x, u=u, xc=xc, A=A.cpu().numpy(), B=B.cpu().numpy(), C=C.cpu().numpy())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variants = {"split": {}, "split_norecur": {"NMPC_EMBED_RECUR": "0"}, "f32": {"NMPC_EMBED": "f32"}, "wide": {"NMPC_EMBED": "wide"}}
    out = {}
    for name, envadd in variants.items():
        env = {k: v for k, v in os.environ.items() if not k.startswith("NMPC_EMBED")}
        env.update(envadd)
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out[name] = np.load(f)
    ref = out["split"]
    sub = slice(0, None, 8)
    x, u, xc = (ref[k].astype(np.float64)[sub] for k in ("x", "u", "xc"))
    Ao, Bo, Co = olpv.embed(omodel, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
    scale = {"A": np.abs(Ao).max(), "B": np.abs(Bo).max(), "C": np.abs(Co).max()}
    worst = {}
    for name, d in out.items():
        for k, o in (("A", Ao), ("B", Bo), ("C", Co)):
            e = np.abs(d[k].astype(np.float64)[sub].reshape(o.shape) - o).max() / scale[k]
            worst[(name, k)] = e
            assert e < 1e-6, (name, k, e)
    for name, d in out.items():
        for k in ("A", "B", "C"):
            e = np.abs(d[k].astype(np.float64) - ref[k].astype(np.float64)).max() / scale[k]
            assert e < 2e-6, (name, k, e)
    print("worst error against the oracle per variant:", {n: max(v for (m, _), v in worst.items() if m == n) for n in variants})


@pytest.mark.parametrize("nx,H", [(2, 16), (2, 32), (3, 48), (4, 32), (2, 65), (2, 96), (8, 65)])
def test_hidden_widths_off_the_kernel_sizes(nx, H):
    """Hidden widths the kernels are not laid out for: below 64 (the reference trains a 32-wide encoder,
    UnbalancedDisc.ipynb; CasADiExp takes any width, mpcUtil.py:6-31) the nets are padded with zero units to 64 and
    run on the 64-wide kernels, 65..127 are padded to 128 on the panel kernel.  Same bar as every other width;
    f(0,0) / h(0) of the padded model are the oracle's."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(nx, 1, 1, H, seed=13)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    assert enc.hidden == H and np.abs(enc.f0 - om.f0).max() < 1e-12 and np.abs(enc.h0 - om.h0).max() < 1e-12
    rng = np.random.default_rng(H)
    for P in (1, 129):
        x = f32(rng.uniform(-2, 2, (P, nx))); u = f32(rng.uniform(-2, 2, (P, 1))); xc = f32(rng.uniform(-2, 2, (P, nx)))
        lam, w = nmpc_amd.quad_nodes("simpson", 20)
        A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
        Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
        for got, ref in ((A, Ao), (B, Bo), (C, Co)):
            err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
            assert err / np.abs(ref).max() < TOL, (nx, H, P, err / np.abs(ref).max())


@pytest.mark.parametrize("nx,H", [(3, 128), (8, 256)])
def test_panel_kernel_out_of_range_and_huge_bias(nx, H):
    """The panel kernel scales its fp16 operands towards the top of the fp16 range with bounds taken from the weights
    alone (worst-case L1 norms), so no input may overflow them: scheduling points hundreds of standard deviations out
    (every unit saturated somewhere on the lambda-path), a layer-1 bias of +-60 (units pinned unless the input is
    enormous) and ordinary points in the same launch, against the fp64 oracle at the panel kernel's bar."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(nx, 1, 1, H, seed=17)
    om.fn.b0[5] = 60.0; om.fn.b0[H - 3] = -60.0; om.hn.b0[9] = -60.0
    om = Model(om.fn, om.hn, None)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    rng = np.random.default_rng(3)
    P = 96
    x = f32(rng.uniform(-2, 2, (P, nx))); u = f32(rng.uniform(-2, 2, (P, 1))); xc = f32(rng.uniform(-2, 2, (P, nx)))
    x[16:32] *= 400.0; u[16:32] *= 400.0; xc[16:32] *= 400.0          # a whole workgroup far out
    x[40] *= 1500.0; xc[41] *= 1500.0; u[70] *= 3000.0                  # single points among ordinary ones
    x[48:64] *= 60.0; xc[48:64] *= 60.0                                 # lam * W0 z sweeps through the pinned units' biases
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
    assert torch.isfinite(A).all() and torch.isfinite(B).all() and torch.isfinite(C).all()
    Ao, Bo, Co = olpv.embed(om, x, u, *olpv.merge_nodes(*olpv.quad_nodes("simpson", 20)), xc=xc)
    worst = 0.0
    for got, ref in ((A, Ao), (B, Bo), (C, Co)):
        err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert err < 5e-6, err              # (2.4e-6 observed at H = 128; ordinary points: 8e-7; the bar is 1e-5)
    print("panel kernel, nx %d H %d, out-of-range points and pinned units: worst error %.2e of the matrix scale" % (nx, H, worst))


@pytest.mark.parametrize("nx,ny,H", [(5, 1, 64), (6, 2, 64), (7, 1, 48), (8, 2, 64), (8, 1, 32)])
def test_five_to_eight_states_on_the_split_tile(nx, ny, H):
    """Nets with five to eight states, one input and one or two outputs (the reference's sin/cos encoder: nx = 8,
    two outputs, sinCos.ipynb cell 6; deepSI's default width 64) run on the split-fp16 tile as three or four tiles of
    two output rows per 16 points (rows 4..7 from the second block of reverse fragments, nine inputs through the third
    input table, linear branch and origin Jacobian from the [8][12] tables): the bar of every other kernel against the
    fp64 oracle - also far out of range and with a pinned unit - at ragged sizes, and the same launch twice gives
    the same bits."""
    import nmpc_amd
    from oracle.model import Model
    om = Model.synthetic(nx, 1, ny, H, seed=21)
    om.fn.b0[3] = 45.0
    om = Model(om.fn, om.hn, None)
    params = {k: [np.asarray(a, np.float32) for a in n.params()] for k, n in (("fn", om.fn), ("hn", om.hn))}
    enc = nmpc_amd.Encoder(params, device="cuda:0")
    rng = np.random.default_rng(nx * 10 + ny)
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    nodes = olpv.merge_nodes(*olpv.quad_nodes("simpson", 20))
    for P in (1, 16, 17, 200):
        x = f32(rng.uniform(-2, 2, (P, nx))); u = f32(rng.uniform(-2, 2, (P, 1))); xc = f32(rng.uniform(-2, 2, (P, nx)))
        if P == 200:
            x[16:32] *= 300.0; u[16:32] *= 300.0; xc[16:32] *= 300.0; x[77] *= 2000.0; xc[78] *= 2000.0
        A, B, C = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
        assert A.shape == (P, nx, nx) and B.shape == (P, nx, 1) and C.shape == (P, ny, nx)
        assert torch.isfinite(A).all() and torch.isfinite(B).all() and torch.isfinite(C).all()
        Ao, Bo, Co = olpv.embed(om, x, u, *nodes, xc=xc)
        for got, ref in ((A, Ao), (B, Bo), (C, Co)):
            err = np.abs(got.cpu().numpy().astype(np.float64).reshape(ref.shape) - ref).max()
            assert err / np.abs(ref).max() < TOL, (nx, ny, H, P, err / np.abs(ref).max())
        A2, B2, C2 = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), torch.tensor(xc), lam, w)
        assert torch.equal(A, A2) and torch.equal(B, B2) and torch.equal(C, C2)
        # A and B only, C only
        A3, B3, _ = nmpc_amd.embed(enc, torch.tensor(x), torch.tensor(u), None, lam, w)
        _, _, C3 = nmpc_amd.embed(enc, None, None, torch.tensor(xc), lam, w)
        assert torch.equal(A, A3) and torch.equal(B, B3) and torch.equal(C, C3)
