"""CPU tests of the reference-signal generator and the run statistics (Reference.py:9-20,
CompTime.ipynb cells 2-5) against the reference's own recorded experiment logs."""
import os
import random

import numpy as np

from conftest import GOLDEN


def test_comp_time_stats_on_reference_logs():
    """Experiments/components_time.npy and lpv_counter.npy of the reference (Nsim 450, 4 passes max):
    the statistics of its CompTime notebook."""
    import nmpc_amd
    ct = np.load(os.path.join(GOLDEN, "ref_components_time.npy"))
    lc = np.load(os.path.join(GOLDEN, "ref_lpv_counter.npy"))
    mx, top5, mean, std, solver = nmpc_amd.comp_time_stats(ct, 450)
    # cell 5 of the notebook, evaluated literally
    groups = np.split(ct, 450, axis=1)
    T = np.array([np.sum(np.sum(g, axis=1)) for g in groups]); S = np.array([np.sum(g, axis=1)[1] for g in groups])
    srt = np.sort(T)
    want = (np.max(T) * 1000, np.mean(srt[int(450 * 0.95):]) * 1000, np.mean(T) * 1000, np.std(T) * 1000, np.mean(S) * 1000)
    assert np.allclose((mx, top5, mean, std, solver), want, rtol=1e-12)
    assert abs(mean - 25.57) < 0.01 and abs(solver - 4.46) < 0.01
    assert nmpc_amd.lpv_histogram(lc) == {1: 338, 2: 110, 3: 2}


def test_random_level_reference():
    import nmpc_amd
    from oracle.plant import random_level_reference
    random.seed(3)
    r = nmpc_amd.randomLevelReference(200, [10, 15], [-1, 1])
    assert r.shape == (200,) and np.all(np.abs(r * 10 - np.round(r * 10)) < 1e-9) and r.min() >= -1 and r.max() <= 1
    runs = np.diff(np.flatnonzero(np.diff(r) != 0))
    assert runs.min() >= 10                       # merged equal neighbours may exceed 15
    # the seeded NumPy form is the benchmark's reference signal
    a = nmpc_amd.random_level_batch(3, 120, [10, 15], [-1, 1], seed0=1234)
    for b in range(3):
        assert np.array_equal(a[b], random_level_reference(120, [10, 15], [-1, 1], np.random.default_rng(1234 + b)))
    assert nmpc_amd.stackReferences((np.zeros(5), np.ones(5))).shape == (2, 5)
