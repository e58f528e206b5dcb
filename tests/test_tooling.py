"""CPU tests of the reference-signal generator and the run statistics (Reference.py:9-20,
CompTime.ipynb cells 2-5) against the reference's own recorded experiment logs."""
import os
import random

import numpy as np
import pytest

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


def test_hand_placed_wait_states_of_the_embedding_kernels():
    """The v_fma_mix splits of embed.hip are inline asm, which the compiler's hazard recognizer does not see; an
    MFMA needs two wait states after a VALU write of an operand.  The checker disassembles every instantiation in
    the built object; it must itself flag a fabricated violation."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_hazards
    fake = ("0000000000001000 <k>:\n"
            "\tv_fma_mixhi_f16 v12, v3, -1.0, v20 op_sel:[1,0,0] // 000000001000: 00\n"
            "\ts_nop 0 // 000000001008: 00\n"
            "\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[12:15], v[0:3] // 00000000100C: 00\n"
            "\tv_fma_mixlo_f16 v30, v3, -1.0, v20 // 000000001014: 00\n"
            "\ts_nop 1 // 000000001018: 00\n"
            "\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[28:31], v[0:3] // 00000000101C: 00\n")
    k, nw, nm, findings = check_hazards.check(fake)
    assert (k, nw, nm) == (1, 2, 2) and len(findings) == 1 and "v[12:15]" in findings[0]
    for name in ("embed.o", "embed_panel.o"):         # both files that split operands in inline asm
        obj = os.path.join(root, "nmpc_amd", "csrc", "build", name)
        if not (os.path.exists(obj) and os.path.exists(check_hazards.OBJDUMP)):
            pytest.skip(name + " not built here")
        k, nw, nm, findings = check_hazards.check(check_hazards.disassemble(obj))
        assert k >= 6 and nw > 100 and nm > 1000          # every instantiation was seen
        assert not findings, findings[:5]
