"""The allocation-free target selector (nmpc_target_select_static) against nmpc_target_select, called
through the C ABI: identical results bit for bit on two calls in a row, the second with a smaller input
(the static scratch still holds the first call's contents), a clear error beyond the static size, and the property the
entry point exists for: a warmed-up call changes neither the device's free memory nor waits for the device."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import disc_setup

pytestmark = pytest.mark.gpu


def _call(fn, enc, yrn, nsamp, s, lam, w, max_lpv):
    from nmpc_amd import _lib
    nb = yrn.shape[0]
    dev = yrn.device
    Xs = torch.full((nb, nsamp, 2), float("nan"), dtype=torch.float64, device=dev)
    Us = torch.full((nb, nsamp), float("nan"), dtype=torch.float64, device=dev)
    Es = torch.full((nb, nsamp), float("nan"), dtype=torch.float64, device=dev)
    it = torch.full((nb, nsamp), -1, dtype=torch.int32, device=dev)
    st = torch.full((nb,), -1, dtype=torch.int32, device=dev)
    lam = np.ascontiguousarray(lam, np.float32); w = np.ascontiguousarray(w, np.float32)
    rc = fn(enc.handle, nb, nsamp, _lib.ptr(yrn), float(s["umin"]), float(s["umax"]), float(s["ymin"][0]), float(s["ymax"][0]),
            lam.ctypes.data_as(_lib.c_float_p), w.ctypes.data_as(_lib.c_float_p), lam.size, max_lpv, 1e-4,
            _lib.ptr(Xs), _lib.ptr(Us), _lib.ptr(Es), _lib.ptr(it), _lib.ptr(st), _lib.current_stream())
    return rc, (Xs, Us, Es, it, st)


def test_static_target_selector_matches_allocating_one(enc, omodel):
    import nmpc_amd
    from nmpc_amd import _lib
    from oracle.plant import random_level_reference
    L = _lib.lib()
    s = disc_setup(omodel)
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    torch.cuda.set_device(0)
    for nb, nsamp, max_lpv in ((300, 40, 20), (37, 25, 20), (64, 30, 0)):       # second call smaller than the first
        refs = np.stack([random_level_reference(nsamp, [10, 15], [-1, 1], np.random.default_rng(500 + b)) for b in range(nb)])
        refs[1] *= 1.4                                                             # clipped at the output bound
        yrn = torch.as_tensor(np.clip((refs - omodel.y0) / omodel.ystd, s["ymin"], s["ymax"]), device="cuda:0").contiguous()
        rc0, ref = _call(L.nmpc_target_select, enc, yrn, nsamp, s, lam, w, max_lpv)
        rc1, got = _call(L.nmpc_target_select_static, enc, yrn, nsamp, s, lam, w, max_lpv)
        assert rc0 == 0 and rc1 == 0, L.nmpc_last_error().decode()
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        assert not torch.isnan(got[0]).any() and int(got[4].abs().max()) == 0 and int(got[3].min()) >= 1


def test_static_target_selector_rejects_oversize(enc, omodel):
    from nmpc_amd import _lib
    import nmpc_amd
    L = _lib.lib()
    s = disc_setup(omodel)
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    nb = 60000                                      # 76 B of scratch per controller at nx = 2: above the 4 MiB array
    assert L.nmpc_target_workspace_bytes(2, nb, 20) > (4 << 20)
    yrn = torch.zeros(nb, 3, dtype=torch.float64, device="cuda:0")
    rc, _ = _call(L.nmpc_target_select_static, enc, yrn, 3, s, lam, w, 20)
    assert rc != 0 and "static buffer" in L.nmpc_last_error().decode()
    torch.cuda.synchronize()


def test_static_target_selector_allocates_nothing_and_does_not_wait(enc, omodel):
    """After one warm-up call (first use sets kernel attributes and looks the scratch symbol up) a call must leave the
    device's free memory unchanged - read through hipMemGetInfo, which sees hipMalloc / hipFree of any caller in the
    process, torch's caching allocator included, because all output tensors are created before the first reading -
    and must be asynchronous: two calls issued on a side stream behind a long-running kernel return before that kernel
    has finished, and their results are those of the allocating entry point once the stream has been synchronised."""
    import nmpc_amd
    from nmpc_amd import _lib
    from oracle.plant import random_level_reference
    L = _lib.lib()
    s = disc_setup(omodel)
    lam, w = nmpc_amd.quad_nodes("simpson", 20)
    torch.cuda.set_device(0)
    nb, nsamp = 128, 30
    refs = np.stack([random_level_reference(nsamp, [10, 15], [-1, 1], np.random.default_rng(900 + b)) for b in range(nb)])
    yrn = torch.as_tensor(np.clip((refs - omodel.y0) / omodel.ystd, s["ymin"], s["ymax"]), device="cuda:0").contiguous()
    rc, ref = _call(L.nmpc_target_select, enc, yrn, nsamp, s, lam, w, 20)
    assert rc == 0
    side = torch.cuda.Stream()
    spin = torch.empty(1 << 26, device="cuda:0")         # work for the side stream to queue behind
    done = torch.cuda.Event()
    with torch.cuda.stream(side):                        # warm-up on the stream that is measured: the HIP runtime
        spin.mul_(1.0000001)                             # itself takes device memory for a stream's first launches
        rc, _ = _call(L.nmpc_target_select_static, enc, yrn, nsamp, s, lam, w, 20)
        assert rc == 0
        done.record(side)
    torch.cuda.synchronize()
    # every tensor the calls need exists before the first reading of the free memory
    bufs = [_alloc_outputs(nb, nsamp, yrn.device) for _ in range(2)]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    with torch.cuda.stream(side):
        for _ in range(40):
            spin.mul_(1.0000001)
        for b in bufs:
            rc = _call_into(L.nmpc_target_select_static, enc, yrn, nsamp, s, lam, w, 20, b)
            assert rc == 0, L.nmpc_last_error().decode()
        done.record(side)
        returned_before_the_stream_drained = not done.query()
    free1 = torch.cuda.mem_get_info(0)[0]
    side.synchronize()
    assert free1 == free0, "device free memory changed by %d bytes across two static calls" % (free0 - free1)
    assert returned_before_the_stream_drained, "the static entry point waited for the stream"
    for b in bufs:
        for a, r in zip(b, ref):
            assert torch.equal(a, r)


def _alloc_outputs(nb, nsamp, dev):
    return (torch.full((nb, nsamp, 2), float("nan"), dtype=torch.float64, device=dev),
            torch.full((nb, nsamp), float("nan"), dtype=torch.float64, device=dev),
            torch.full((nb, nsamp), float("nan"), dtype=torch.float64, device=dev),
            torch.full((nb, nsamp), -1, dtype=torch.int32, device=dev), torch.full((nb,), -1, dtype=torch.int32, device=dev))


def _call_into(fn, enc, yrn, nsamp, s, lam, w, max_lpv, out):
    from nmpc_amd import _lib
    Xs, Us, Es, it, st = out
    lam = np.ascontiguousarray(lam, np.float32); w = np.ascontiguousarray(w, np.float32)
    return fn(enc.handle, yrn.shape[0], nsamp, _lib.ptr(yrn), float(s["umin"]), float(s["umax"]), float(s["ymin"][0]),
              float(s["ymax"][0]), lam.ctypes.data_as(_lib.c_float_p), w.ctypes.data_as(_lib.c_float_p), lam.size, max_lpv, 1e-4,
              _lib.ptr(Xs), _lib.ptr(Us), _lib.ptr(Es), _lib.ptr(it), _lib.ptr(st), _lib.current_stream())
