"""Reference-signal generator and run statistics of the reference's tooling
(Reference.py:6-20, CompTime.ipynb cells 2-5) for batched runs."""
import random

import numpy as np


def stackReferences(references_tuple):
    """Reference.py:6-7"""
    return np.vstack(references_tuple)


def randomLevelReference(Nsim, nt_range, level_range, rng=None):
    """Piecewise-constant random levels (Reference.py:9-20): segments of nt_range[0]..nt_range[1]
    samples, each at a level drawn from level_range in steps of 0.1.  rng: a random.Random (the
    reference draws from the `random` module's global generator, the default here), or a NumPy
    Generator (the seeded form the batched benchmark uses)."""
    if rng is None:
        rng = random
    if isinstance(rng, np.random.Generator):
        randint = lambda a, b: int(rng.integers(a, b + 1))
    else:
        randint = rng.randint
    out = np.array([])
    remaining = Nsim
    while True:
        steps = randint(nt_range[0], nt_range[1])
        remaining -= steps
        out = np.hstack((out, np.ones(steps) * randint(int(level_range[0] * 10), int(level_range[1] * 10)) / 10))
        if remaining <= 0:
            return out[:Nsim]


def random_level_batch(B, Nsim, nt_range, level_range, seed0=1234):
    """[B, Nsim] references, controller b seeded with default_rng(seed0 + b)."""
    return np.stack([randomLevelReference(Nsim, nt_range, level_range, np.random.default_rng(seed0 + b)) for b in range(B)])


def lpv_histogram(lpv_counter):
    """np.unique(lpv_counter, return_counts=True) of CompTime.ipynb cell 2, as a dict."""
    v, c = np.unique(np.asarray(lpv_counter.cpu() if hasattr(lpv_counter, "cpu") else lpv_counter), return_counts=True)
    return {int(a): int(b) for a, b in zip(v, c)}


def comp_time_stats(components_time, Nsim):
    """CompTime.ipynb cells 3-5: per-step total T_iter and solver S_iter from the 4 x (Nsim*max_iterations)
    bucket log; returns (max, mean of the slowest 5 %, mean, std of T_iter, mean of S_iter), in ms."""
    groups = np.split(np.asarray(components_time), Nsim, axis=1)
    S = np.array([g.sum(axis=1)[1] for g in groups])
    T = np.array([g.sum() for g in groups])
    srt = np.sort(T)
    return (T.max() * 1e3, srt[int(Nsim * 0.95):].mean() * 1e3, T.mean() * 1e3, T.std() * 1e3, S.mean() * 1e3)
