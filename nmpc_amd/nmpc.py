"""Closed-loop LPV-NMPC controllers with the reference's call signatures, batched over
independent controllers (scenarios / references / noise seeds) on one GPU.

output_NMPC_linear follows NMPC_test.py:22-320: every array the reference returns comes back with
a leading batch dimension (dropped again when a single 1-D reference is passed)."""
import ctypes
import time

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, current_stream
from .lpv import quad_nodes
from .mpcutil import _f64


class OutputUnbalancedDisc:
    """Plant descriptor with the constants of Systems.py:79-97 (angle output + N(0, sigma_n))."""
    g = 9.80155078791343
    J = 0.000244210523960356
    Km = 10.5081817407479
    I = 0.0410772235841364
    M = 0.0761844495320390
    tau = 0.397973147009910

    def __init__(self, dt=0.025, sigma_n=(0,), seed=0):
        self.dt = float(dt)
        self.sigma_n = list(sigma_n)
        self.seed = int(seed)
        self.nx = 2
        self.x0 = np.zeros(2)

    def reset_state(self):
        self.x0 = np.zeros(2)

    def params(self):
        return [self.g, self.J, self.Km, self.I, self.M, self.tau]


class NoisyUnbalancedDisc(OutputUnbalancedDisc):
    """Full-state measurement x + N(0, sigma_n[r]) per component (Systems.py:58-77)."""

    def __init__(self, dt=0.025, sigma_n=(0, 0), seed=0):
        super().__init__(dt, sigma_n, seed)


class ModelPlant:
    """The model as its own plant: x+ = f(x, u), y = h(x) + N(0, sigma_n), the state known to the controller (no
    observer).  For models whose physical plant is not part of this package (state dimensions up to 8, hidden
    width up to 256); pass it where the reference's scripts pass their deepSI system."""

    def __init__(self, dt=0.1, sigma_n=(0,), seed=0):
        self.dt = float(dt)
        self.sigma_n = list(sigma_n)
        self.seed = int(seed)
        self.x0 = None

    def reset_state(self):
        self.x0 = None

    def params(self):
        return [0.0] * 6


def mpc_qp(list_A, list_B, list_C, x0, y_meas, Xs, Us, f0, h0, Q, R, rho, u_min, u_max, y_min, y_max, Nc, nx,
           want_matrices=False):
    """One condensed QP of the output-constrained controller (NMPC_test.py:206-238), batched.
    Returns dict(u, eps, xpred, status[, Ge, Fe, Le, h])."""
    lA = _f64(list_A); dev = lA.device
    batch = lA.shape[:-2]
    nb = int(np.prod(batch)) if len(batch) else 1
    lB = _f64(list_B, dev).reshape(batch + (Nc * nx,)); lC = _f64(list_C, dev).reshape(batch + (Nc, nx))
    x0 = _f64(x0, dev).reshape(batch + (nx,)); ym = _f64(y_meas, dev).reshape(batch)
    Xs = _f64(Xs, dev).reshape(batch + (Nc * nx,)); Us = _f64(Us, dev).reshape(batch + (Nc,))
    n, m = Nc + 1, 4 * Nc + 2
    mk = lambda *s: torch.empty(batch + s, dtype=torch.float64, device=dev)
    out = dict(u=mk(Nc), eps=mk(), xpred=mk(Nc * nx), status=torch.empty(batch, dtype=torch.int32, device=dev))
    if want_matrices:
        out.update(Ge=mk(n, n), Fe=mk(n), Le=mk(m, n), h=mk(m))
    f0a = np.ascontiguousarray(np.asarray(f0, np.float64).reshape(-1))
    Qa = np.ascontiguousarray(np.asarray(Q, np.float64).reshape(nx, nx))
    R = float(np.asarray(R).reshape(-1)[0])
    with torch.cuda.device(dev):
        rc = lib().nmpc_mpc_qp(nb, Nc, nx, ptr(lA), ptr(lB), ptr(lC), ptr(x0), ptr(ym), ptr(Xs), ptr(Us),
                               f0a.ctypes.data_as(_lib.c_double_p), float(np.asarray(h0).reshape(-1)[0]),
                               Qa.ctypes.data_as(_lib.c_double_p), R, float(rho), float(u_min), float(u_max),
                               float(y_min), float(y_max), ptr(out.get("Ge")), ptr(out.get("Fe")),
                               ptr(out.get("Le")), ptr(out.get("h")), ptr(out["u"]), ptr(out["eps"]),
                               ptr(out["xpred"]), ptr(out["status"]), current_stream())
    check(rc, "nmpc_mpc_qp")
    return out


class Controller:
    """Batch of closed-loop controllers resident on the GPU (nmpc_ctrl of the C ABI)."""

    def __init__(self, encoder, system, batch, Nc, Nsim, max_iterations, Q, R, u_min_n, u_max_n, y_min_n, y_max_n,
                 nodes, weights, tol=1e-1, rho=1000.0, state_feedback=None):
        """state_feedback: None for the output-feedback controller, or a dict(x_min, x_max [nx] normalised
        state bounds, y0, ystd [nx] state normalisation, hard=bool, as_written=bool) for the state-feedback
        controllers (as_written: input targets and linear cost term as NMPC.py:580,591 have them)."""
        model_plant = isinstance(system, ModelPlant)
        if encoder.nx not in (2, 3, 4, 5, 6, 7, 8):
            raise _lib.NmpcError("closed-loop controllers support nx = 2..8 (model has nx = %d)" % encoder.nx)
        cfg = _lib.CtrlConfig()
        cfg.batch, cfg.Nc, cfg.max_iterations, cfg.n_sim = int(batch), int(Nc), int(max_iterations), int(Nsim)
        cfg.tol, cfg.rho = float(tol), float(rho)
        Qa = np.asarray(Q, np.float64).reshape(encoder.nx, encoder.nx)
        for i, v in enumerate(Qa.reshape(-1)):
            cfg.Q[i] = float(v)
        cfg.R = float(np.asarray(R).reshape(-1)[0])
        cfg.u_min, cfg.u_max, cfg.y_min, cfg.y_max = float(u_min_n), float(u_max_n), float(y_min_n), float(y_max_n)
        nm = encoder.norm
        cfg.u0, cfg.ustd, cfg.y0, cfg.ystd = nm.u0, nm.ustd, nm.y0, nm.ystd
        cfg.dt = float(system.dt)
        for i, v in enumerate(system.params()):
            cfg.plant[i] = float(v)
        if not 1 <= len(nodes) <= len(cfg.lam) or len(weights) != len(nodes):
            raise _lib.NmpcError("need 1..%d quadrature nodes with as many weights (got %d, %d)"
                                 % (len(cfg.lam), len(nodes), len(weights)))
        cfg.n_nodes = len(nodes)
        for i in range(len(nodes)):
            cfg.lam[i] = float(nodes[i]); cfg.w[i] = float(weights[i])
        self.state_feedback = state_feedback is not None
        if model_plant:
            cfg.mode = 2
        elif state_feedback is None:         # NMPC_test.py:115-118: uhist has encoder.na entries, yhist encoder.nb + 1
            cfg.n_uhist, cfg.n_yhist = int(encoder.na), int(encoder.nb) + 1
        if state_feedback is not None:
            cfg.mode, cfg.hard = 1, int(bool(state_feedback.get("hard", False)))
            cfg.as_written = int(bool(state_feedback.get("as_written", False)))
            for i in range(encoder.nx):
                cfg.x_min[i] = float(np.asarray(state_feedback["x_min"]).reshape(-1)[i])
                cfg.x_max[i] = float(np.asarray(state_feedback["x_max"]).reshape(-1)[i])
                cfg.x_norm0[i] = float(np.asarray(state_feedback["y0"]).reshape(-1)[i])
                cfg.x_normstd[i] = float(np.asarray(state_feedback["ystd"]).reshape(-1)[i])
        self.encoder, self.cfg = encoder, cfg
        self.B, self.Nc, self.Nsim, self.nx = int(batch), int(Nc), int(Nsim), encoder.nx
        h = ctypes.c_void_p()
        with torch.cuda.device(encoder.device):
            rc = lib().nmpc_ctrl_create(ctypes.byref(h), encoder.handle, ctypes.byref(cfg))
        check(rc, "nmpc_ctrl_create")
        self.handle = h
        self.Xs = self.Us = self.noise = None
        dev = encoder.device
        z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=dev)
        B, Ns = self.B, self.Nsim
        self.logs = dict(x_log=z(B, Ns + 1, self.nx), u_log=z(B, Ns), e_log=z(B, Ns), y_log=z(B, Ns + 1),
                         y_est_log=z(B, Ns + 1), lpv_counter=z(B, Ns, dt=torch.int32), plan_log=z(B, Ns, Nc),
                         qp_status=z(B, Ns, dt=torch.int8))
        check(lib().nmpc_ctrl_set_status_log(self.handle, ptr(self.logs["qp_status"])), "nmpc_ctrl_set_status_log")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                lib().nmpc_ctrl_destroy(h)
            except Exception:
                pass
            self.handle = None

    def reset(self, x0n, u0n, plant_x0=None):
        x0a = np.ascontiguousarray(np.asarray(x0n, np.float64).reshape(-1))
        px = None if plant_x0 is None else np.ascontiguousarray(np.asarray(plant_x0, np.float64).reshape(-1))
        with torch.cuda.device(self.encoder.device):
            rc = lib().nmpc_ctrl_reset(self.handle, x0a.ctypes.data_as(_lib.c_double_p), float(u0n),
                                       None if px is None else px.ctypes.data_as(_lib.c_double_p), current_stream())
        check(rc, "nmpc_ctrl_reset")

    def set_targets(self, Xs, Us, noise=None):
        """Xs [B, Nsim+Nc, nx], Us [B, Nsim+Nc], noise [Nsim, B] (state feedback: [Nsim, B, nx]), device fp64."""
        dev = self.encoder.device
        self.Xs = _f64(Xs, dev).reshape(self.B, self.Nsim + self.Nc, self.nx)
        self.Us = _f64(Us, dev).reshape(self.B, self.Nsim + self.Nc)
        shape = (self.Nsim, self.B, self.nx) if self.state_feedback else (self.Nsim, self.B)
        self.noise = None if noise is None else _f64(noise, dev).reshape(shape)

    def run(self, k0, nsteps):
        if self.Xs is None:
            raise _lib.NmpcError("Controller.run: set_targets() has not been called")
        L = self.logs
        with torch.cuda.device(self.encoder.device):
            rc = lib().nmpc_ctrl_run(self.handle, int(k0), int(nsteps), ptr(self.Xs), ptr(self.Us), ptr(self.noise),
                                     ptr(L["x_log"]), ptr(L["u_log"]), ptr(L["e_log"]), ptr(L["y_log"]),
                                     ptr(L["y_est_log"]), ptr(L["lpv_counter"]), ptr(L["plan_log"]), current_stream())
        check(rc, "nmpc_ctrl_run")

    def profile(self, enable=True):
        """Bracket the kernel launches of later run() calls with HIP events (see profile_read):
        True / 1 every step, n > 1 the steps k with k % n == 0, False / 0 none."""
        check(lib().nmpc_ctrl_profile(self.handle, int(enable)), "nmpc_ctrl_profile")

    def profile_read(self):
        """{kind: (launches, total_ms)} for kinds embed / solve / plant since the last read."""
        n = (ctypes.c_int * 3)(); ms = (ctypes.c_double * 3)()
        with torch.cuda.device(self.encoder.device):
            check(lib().nmpc_ctrl_profile_read(self.handle, n, ms, current_stream()), "nmpc_ctrl_profile_read")
        return {k: (int(n[i]), float(ms[i])) for i, k in enumerate(("embed", "solve", "plant"))}

    def embed_clock(self, enable=True, slots=8192):
        """Diagnostics: enable=True makes the first-pass embedding launches of later run() calls stamp their
        workgroups' lifetimes in shader-clock cycles and 100 MHz ticks; enable=False stops that and returns the
        median clock in GHz over the workgroups stamped since (None if none were)."""
        if enable:
            self._stamps = torch.zeros(slots, 2, dtype=torch.int64, device=self.encoder.device)
            check(lib().nmpc_ctrl_set_embed_stamps(self.handle, ptr(self._stamps), slots), "nmpc_ctrl_set_embed_stamps")
            return None
        check(lib().nmpc_ctrl_set_embed_stamps(self.handle, None, 0), "nmpc_ctrl_set_embed_stamps")
        st = getattr(self, "_stamps", None)
        if st is None:
            return None
        torch.cuda.synchronize(self.encoder.device)
        v = st.cpu().numpy().astype(np.float64)
        v = v[v[:, 1] > 0]
        self._stamps = None
        return float(np.median(v[:, 0] / v[:, 1]) * 0.1) if len(v) else None

    @property
    def tail_from(self):
        """First LPV pass (0-based) that runs in the per-controller tail launch; 0 if every pass is a batch-wide launch."""
        return int(lib().nmpc_ctrl_tail_from(self.handle))

    def profile_read_steps(self, k0, nsteps):
        """[nsteps, 3] ms per step and kind (embed, solve, plant) since the last read."""
        ms = np.zeros((nsteps, 3))
        with torch.cuda.device(self.encoder.device):
            check(lib().nmpc_ctrl_profile_read_steps(self.handle, int(k0), int(nsteps),
                                                     ms.ctypes.data_as(_lib.c_double_p), current_stream()),
                  "nmpc_ctrl_profile_read_steps")
        return ms

    def stats(self):
        """(failed QP solves, QP solves, active-set iterations) since reset; synchronises."""
        v = (ctypes.c_longlong * 4)()
        with torch.cuda.device(self.encoder.device):
            check(lib().nmpc_ctrl_stats(self.handle, v, current_stream()), "nmpc_ctrl_stats")
        self.slow_solves = int(v[3])        # long horizons: solves the first-stage kernel handed to the full kernel
        return int(v[0]), int(v[1]), int(v[2])

    def qp_failures(self):
        return self.stats()[0]

    def failed_mask(self):
        """[B, Nsim] bool: controller b had a QP solve fail in step k (it then held its previous plan, shifted)."""
        return self.logs["qp_status"] != 0


def _report_failures(ctrl, what, on_qp_failure):
    """Per-controller handling of failed QP solves (the reference would stop at NMPC_test.py:233-235 with the
    first one): the loop has already kept every other controller going; say so, or raise if asked to."""
    nfail = ctrl.qp_failures()
    if not nfail or on_qp_failure == "ignore":
        return
    mask = ctrl.failed_mask()
    msg = ("%s: %d QP solves failed in %d of %d controllers (first: controller %d); those controllers held their "
           "previous plan in the steps concerned, see the status log"
           % (what, nfail, int(mask.any(1).sum()), mask.shape[0], int(mask.any(1).nonzero()[0, 0]) if mask.any() else -1))
    if on_qp_failure == "raise":
        raise _lib.NmpcError(msg)
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


_PROFILE_CHUNK = 256


def _timed_loop(ctrl, dev, Nsim):
    """Runs Nsim closed-loop steps one call at a time with an event after each and the per-launch event pairs of
    the controller switched on; the pairs are read back (and their events recycled) every _PROFILE_CHUNK steps, so
    that a long run holds a bounded number of live events.  Returns (per-step seconds [Nsim], per-step kernel ms
    [Nsim, 3], wall seconds)."""
    comp_t = np.zeros(Nsim)
    kms = np.zeros((Nsim, 3))
    torch.cuda.synchronize(dev)
    start = time.time()
    ctrl.profile(1)
    for k0 in range(0, Nsim, _PROFILE_CHUNK):
        n = min(_PROFILE_CHUNK, Nsim - k0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for k in range(n):
            ctrl.run(k0 + k, 1)
            ev[k + 1].record()
        kms[k0:k0 + n] = ctrl.profile_read_steps(k0, n)          # synchronises the stream
        for k in range(n):
            comp_t[k0 + k] = ev[k].elapsed_time(ev[k + 1]) * 1e-3
    torch.cuda.synchronize(dev)
    runtime = time.time() - start
    ctrl.profile(0)
    return comp_t, kms, runtime


def output_NMPC_linear(system, encoder, u_min, u_max, y_min, y_max, x0, Q, R, dt, dlam, stages, y_reference_list,
                       Nc=5, Nsim=30, max_iterations=1, *, tol=1e-1, rho=1000.0, targets=None, noise=None,
                       return_controller=False, return_status=False, on_qp_failure="warn"):
    """Output-constrained LPV-NMPC with observer (NMPC_test.py:22-320), batched.

    y_reference_list: [B, >= Nsim+Nc] (or 1-D for one controller) output reference in physical units.
    targets: optional (Xs [B, nx, Nsim+Nc], Us [B, nu, Nsim+Nc]) steady-state targets; computed by
    getXsUs_Cs when omitted.  noise: optional [Nsim, B] measurement noise; drawn from
    N(0, system.sigma_n[0]) with system.seed when omitted and sigma_n > 0.
    Returns the reference's 12-tuple (x_log, u_log, e_log, comp_t_log, t, runtime, lpv_counter,
    components_time, y_log, y_est_log, Xs, Us); with return_status=True a 13th entry, the int8 status log
    [B, Nsim] (0 = every QP solve of that controller's step succeeded).  on_qp_failure: "warn" (default), "raise"
    or "ignore" - a controller whose solve fails holds its previous plan for that step, the others run on."""
    _lib.require_gpu()
    dev = encoder.device
    nx, nu, ny = encoder.nx, encoder.nu, encoder.ny
    nm = encoder.norm
    yref = torch.as_tensor(np.asarray(y_reference_list, np.float64))
    single = yref.dim() == 1
    if single:
        yref = yref[None]
    B = yref.shape[0]
    assert yref.shape[1] >= Nsim + Nc, "reference must hold Nsim+Nc samples"
    y_ref_n = ((yref - nm.y0) / nm.ystd)                                  # NMPC_test.py:47
    x0 = np.asarray(x0, np.float64).reshape(nx)
    x0n = (x0 - nm.y0) / nm.ystd                                          # NMPC_test.py:48
    u0n = (0 - nm.u0) / nm.ustd                                           # NMPC_test.py:49-50
    ymax_n = (float(np.asarray(y_max).reshape(-1)[0]) - nm.y0) / nm.ystd  # NMPC_test.py:70-73
    ymin_n = (float(np.asarray(y_min).reshape(-1)[0]) - nm.y0) / nm.ystd
    umin_n = (float(u_min) - nm.u0) / nm.ustd
    umax_n = (float(u_max) - nm.u0) / nm.ustd
    y_ref_n = torch.clamp(y_ref_n, ymin_n, ymax_n)                        # NMPC_test.py:76
    lam, w = quad_nodes("simpson", stages)                                # lambda_simpson, NMPC_test.py:59
    if targets is None:
        from .target import getXsUs_Cs
        from .target import GetABC
        Xs, Us, _ = getXsUs_Cs(y_ref_n[:, :Nsim + Nc].to(dev), nx, nu, ny, Nsim + Nc, umin_n, umax_n, [ymin_n],
                               [ymax_n], GetABC(encoder, lam, w))
    else:
        Xs, Us = (_f64(t, dev) for t in targets)
        Xs = Xs.reshape(B, nx, -1)[:, :, :Nsim + Nc]; Us = Us.reshape(B, nu, -1)[:, :, :Nsim + Nc]
    ctrl = Controller(encoder, system, B, Nc, Nsim, max_iterations, Q, R, umin_n, umax_n, ymin_n, ymax_n, lam, w,
                      tol=tol, rho=rho)
    if noise is None and system.sigma_n[0] > 0:
        gen = torch.Generator(device="cpu"); gen.manual_seed(system.seed)
        noise = torch.randn(Nsim, B, generator=gen, dtype=torch.float64) * float(system.sigma_n[0])
    ctrl.set_targets(Xs.transpose(1, 2).contiguous(), Us[:, 0, :].contiguous(), noise)
    ctrl.reset(x0n, u0n, getattr(system, "x0", None))
    comp_t_log, kms, runtime = _timed_loop(ctrl, dev, Nsim)
    kms = kms * 1e-3
    _report_failures(ctrl, "output_NMPC_linear", on_qp_failure)
    L = ctrl.logs
    x_log = L["x_log"].clone(); x_log[:, 0, :] = torch.as_tensor(x0, device=dev)   # NMPC_test.py:88
    x_log = x_log.reshape(B, (Nsim + 1) * nx)
    e_log = torch.zeros(B, nx, Nsim, dtype=torch.float64, device=dev); e_log[:, 0, :] = L["e_log"]
    t = np.arange(Nsim) * dt
    # buckets of the reference (NMPC_test.py:152-300): getAB, solve, overhead, sim.  Column
    # k*max_iterations holds step k, so that the reference's analysis (split into Nsim groups of
    # max_iterations columns and sum, CompTime.ipynb cells 3-4) recovers per-step totals.
    components_time = np.zeros((4, Nsim * max_iterations))
    components_time[0, ::max_iterations] = kms[:, 0]
    components_time[1, ::max_iterations] = kms[:, 1]
    components_time[3, ::max_iterations] = kms[:, 2]
    components_time[2, ::max_iterations] = np.maximum(comp_t_log - kms.sum(1), 0.0)
    out = [x_log, L["u_log"], e_log, comp_t_log, t, runtime, L["lpv_counter"], components_time, L["y_log"],
           L["y_est_log"], Xs, Us]
    if return_status:
        out.append(L["qp_status"])
    if single:
        out = [o[0] if torch.is_tensor(o) else o for o in out]
    if return_controller:
        return tuple(out), ctrl
    return tuple(out)


def _state_loop(system, encoder, x_min, x_max, u_min, u_max, x0, Q, R, dt, dlam, stages, x_reference_list, Nc, Nsim,
                max_iterations, tol, rho, hard, noise, C, targets, as_written=False, on_qp_failure="warn", what="state-feedback NMPC"):
    """Shared body of par_NMPC_linear / NMPC_linear: returns (x_log, u_log, e_log, comp_t_log, t, runtime,
    lpv_counter, reference_list_normalized, per-step kernel ms [Nsim, 3], controller)."""
    from .target import getXsUs, GetABC
    _lib.require_gpu()
    dev = encoder.device
    nx, nu = encoder.nx, encoder.nu
    nm = encoder.norm
    y0 = np.broadcast_to(np.asarray(nm.y0v if hasattr(nm, "y0v") else nm.y0, np.float64), (nx,)).copy()
    ystd = np.broadcast_to(np.asarray(nm.ystdv if hasattr(nm, "ystdv") else nm.ystd, np.float64), (nx,)).copy()
    lam, w = quad_nodes("simpson", stages, dlam)                       # lpv_int, NMPC.py:400,533
    x0 = np.asarray(x0, np.float64).reshape(-1)
    xmax_n = (np.asarray(x_max, np.float64) - y0) / ystd; xmin_n = (np.asarray(x_min, np.float64) - y0) / ystd
    umin_n = (float(u_min) - nm.u0) / nm.ustd; umax_n = (float(u_max) - nm.u0) / nm.ustd
    refn = None
    if x_reference_list is not None:
        ref = torch.as_tensor(np.asarray(x_reference_list, np.float64))
        single = ref.dim() == 2
        if single:
            ref = ref[None]
        B = ref.shape[0]
        assert ref.shape[1] == nx and ref.shape[2] >= Nsim + Nc, "reference must be [B, nx, >= Nsim+Nc]"
        refn = (ref - torch.as_tensor(y0)[None, :, None]) / torch.as_tensor(ystd)[None, :, None]     # NMPC.py:524
        if targets is None:
            Xs, Us, _ = getXsUs(refn[:, 1, :Nsim + Nc].to(dev), nx, nu, 1, Nsim + Nc, umin_n, umax_n, xmin_n, xmax_n,
                                GetABC(encoder, lam, w), None, np.asarray(C, np.float64), None, 0.0)
        else:
            Xs, Us = (_f64(t, dev) for t in targets)
            Xs = Xs.reshape(B, nx, -1)[:, :, :Nsim + Nc]; Us = Us.reshape(B, nu, -1)[:, :, :Nsim + Nc]
        u0n = (0 - nm.u0) / nm.ustd                                    # NMPC.py:526-527
    else:
        x0 = x0.reshape(-1, nx)
        B, single = x0.shape[0], x0.shape[0] == 1
        Xs = torch.zeros(B, nx, Nsim + Nc, dtype=torch.float64, device=dev)
        Us = torch.zeros(B, nu, Nsim + Nc, dtype=torch.float64, device=dev)
        u0n = 0.0                                                      # NMPC.py:427
    x0b = np.broadcast_to(x0.reshape(-1, nx), (B, nx))
    if not np.all(x0b == x0b[0]):
        raise _lib.NmpcError("state-feedback loops start every controller of a batch from the same x0")
    ctrl = Controller(encoder, system, B, Nc, Nsim, max_iterations, Q, R, umin_n, umax_n, 0.0, 0.0, lam, w, tol=tol,
                      rho=rho, state_feedback=dict(x_min=xmin_n, x_max=xmax_n, y0=y0, ystd=ystd, hard=hard,
                                                   as_written=as_written))
    if noise is None and any(float(sg) > 0 for sg in system.sigma_n):
        gen = torch.Generator(device="cpu"); gen.manual_seed(system.seed)
        noise = torch.randn(Nsim, B, nx, generator=gen, dtype=torch.float64) * torch.as_tensor(
            np.asarray(system.sigma_n, np.float64)[:nx])
    ctrl.set_targets(Xs.transpose(1, 2).contiguous(), Us[:, 0, :].contiguous(), noise)
    ctrl.reset((x0b[0] - y0) / ystd, u0n, x0b[0])
    comp_t_log, kms, runtime = _timed_loop(ctrl, dev, Nsim)
    _report_failures(ctrl, what, on_qp_failure)
    L = ctrl.logs
    x_log = L["x_log"].clone(); x_log[:, 0, :] = torch.as_tensor(np.array(x0b[0]), device=dev)
    x_log = x_log.reshape(B, (Nsim + 1) * nx)
    e_log = torch.zeros(B, nx, Nsim, dtype=torch.float64, device=dev); e_log[:, 0, :] = L["e_log"]
    out = [x_log, L["u_log"], e_log, comp_t_log, np.arange(Nsim) * dt, runtime, L["lpv_counter"], refn]
    if single:
        out = [o[0] if torch.is_tensor(o) else o for o in out]
    return out, kms, ctrl


def par_NMPC_linear(system, encoder, x_min, x_max, u_min, u_max, x0, u_ref, Q, R, dt, dlam, stages, x_reference_list,
                    Nc=5, Nsim=30, max_iterations=1, *, tol=1e-5, rho=10000.0, noise=None, C=(0.0, 1.0), targets=None,
                    return_controller=False, as_written=False, return_status=False, on_qp_failure="warn"):
    """State-feedback LPV-NMPC tracking the targets of getXsUs (NMPC.py:507-650), batched:
    x_reference_list [B, nx, >= Nsim+Nc] (or [nx, ...] for one controller), physical units; row 1 is the
    tracked output (NMPC.py:539-540).  Soft state/input box through one slack (weight 10000).  u_ref is
    accepted and unused, as in the reference.  Returns the reference's 9-tuple (x_log, u_log, e_log,
    comp_t_log, t, runtime, lpv_counter, reference_list_normalized, components_time), plus the int8 status log
    [B, Nsim] with return_status=True.
    as_written=False (default): input targets follow the step (Us = us[k : k+Nc]) and the linear cost term is
    2 (Gamma' Omega (Phi x0 - Xs) - Psi Us), as output_NMPC_linear has them for the same step (NMPC_test.py:152,213)
    - the controller the reference's later code converged on.  as_written=True: NMPC.py:580 (Us = us[:Nc] at every
    step) and NMPC.py:591 (+ Psi' Us) to the letter.  Both keep the nu-wide shift of the input warm start."""
    out, kms, ctrl = _state_loop(system, encoder, x_min, x_max, u_min, u_max, x0, Q, R, dt, dlam, stages,
                                 x_reference_list, Nc, Nsim, max_iterations, tol, rho, False, noise, C, targets,
                                 as_written=as_written, on_qp_failure=on_qp_failure, what="par_NMPC_linear")
    components_time = np.zeros((4, Nsim * max_iterations))              # NMPC.py:560; column k*max_iterations = step k
    components_time[0, ::max_iterations] = kms[:, 0] * 1e-3
    components_time[1, ::max_iterations] = kms[:, 1] * 1e-3
    components_time[3, ::max_iterations] = kms[:, 2] * 1e-3
    components_time[2, ::max_iterations] = np.maximum(out[3] - kms.sum(1) * 1e-3, 0.0)
    res = tuple(out) + (components_time,)
    if return_status:
        res = res + (ctrl.logs["qp_status"][0] if out[1].dim() == 1 else ctrl.logs["qp_status"],)
    return (res, ctrl) if return_controller else res


def NMPC_linear(system, encoder, x_min, x_max, u_min, u_max, x0, u_ref, Q, R, dt, dlam, stages, x_reference_list=None,
                Nc=5, Nsim=30, max_iterations=5, *, tol=1e-7, noise=None, return_controller=False, return_status=False,
                on_qp_failure="warn"):
    """State-feedback LPV-NMPC regulating to the origin with hard state/input constraints
    (NMPC.py:374-505; its QP goes to quadprog, the dual active-set method this library implements;
    at most 5 passes, stop 1e-7).  x0 [nx] or [B, nx] (all rows equal).  Returns the reference's 9-tuple
    (x_log, u_log, e_log, comp_t_log, t, runtime, lpv_counter, reference_list_normalized,
    components_total_time[4]), plus the int8 status log with return_status=True.  A controller whose hard-constrained
    QP is infeasible in some step holds its previous plan there (status 2); on_qp_failure "warn" | "raise" | "ignore"."""
    out, kms, ctrl = _state_loop(system, encoder, x_min, x_max, u_min, u_max, x0, Q, R, dt, dlam, stages, None, Nc,
                                 Nsim, max_iterations, tol, 1.0, True, noise, (0.0, 1.0), None,
                                 on_qp_failure=on_qp_failure, what="NMPC_linear")
    if x_reference_list is not None:                                    # NMPC.py:391 (returned, not tracked)
        nm = encoder.norm
        y0 = np.asarray(nm.y0v if hasattr(nm, "y0v") else nm.y0, np.float64)
        ystd = np.asarray(nm.ystdv if hasattr(nm, "ystdv") else nm.ystd, np.float64)
        out[7] = ((np.asarray(x_reference_list, np.float64).T - y0) / ystd).T
    tot = kms.sum(0) * 1e-3
    components_total_time = np.array([tot[0], tot[1], max(out[3].sum() - tot.sum(), 0.0), tot[2]])   # NMPC.py:423
    res = tuple(out) + (components_total_time,)
    if return_status:
        res = res + (ctrl.logs["qp_status"][0] if out[1].dim() == 1 else ctrl.logs["qp_status"],)
    return (res, ctrl) if return_controller else res
