"""fp64 restatement of the simulated plant and the reference-signal generator.

Plant: unbalanced disc, Systems.py:79-97 (OutputUnbalancedDisc): state (omega, theta),
    d omega = -M g I / J sin(theta) - omega / tau + Km / tau u,   d theta = omega,
output y = theta + N(0, sigma).  Time stepping is one classic RK4 step of length dt per
sample (deepSI System_deriv.f, visible in the traceback of NumpyLinearNMPC.ipynb cell 15:
k1 = dt*deriv(x,u); k2 = dt*deriv(x+k1/2,u); ...).
Reference signal: Reference.py:9-20 (piecewise-constant random levels).
"""
import numpy as np

DISC = dict(g=9.80155078791343, J=0.000244210523960356, Km=10.5081817407479,
            I=0.0410772235841364, M=0.0761844495320390, tau=0.397973147009910)


def disc_deriv(x, u, p=DISC):
    w, th = x[..., 0], x[..., 1]
    dw = -p["M"] * p["g"] * p["I"] / p["J"] * np.sin(th) - 1 / p["tau"] * w + p["Km"] / p["tau"] * u
    return np.stack([dw, w], axis=-1)


def disc_step(x, u, dt, p=DISC):
    k1 = dt * disc_deriv(x, u, p)
    k2 = dt * disc_deriv(x + k1 / 2, u, p)
    k3 = dt * disc_deriv(x + k2 / 2, u, p)
    k4 = dt * disc_deriv(x + k3, u, p)
    return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6


def random_level_reference(Nsim, nt_range, level_range, rng):
    """Reference.py:9-20 with a NumPy Generator in place of the `random` module: segments of
    U{nt0..nt1} samples (inclusive), each at level U{10*lo..10*hi}/10 (inclusive)."""
    out = np.zeros(0)
    remaining = Nsim
    while True:
        steps = int(rng.integers(nt_range[0], nt_range[1] + 1))
        remaining -= steps
        level = int(rng.integers(int(level_range[0] * 10), int(level_range[1] * 10) + 1)) / 10
        out = np.concatenate([out, np.full(steps, level)])
        if remaining <= 0:
            return out[:Nsim]
