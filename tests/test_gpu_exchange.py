"""The multi-device helpers of the C ABI (include/apd_mi355x.h: apd_device_*, apd_rescale_nearest_device, apd_exchange_*) called
directly, on the one GPU of the box: a one-device list through RCCL, lists that repeat the device through the two-level
exchange (RCCL between the leader ranks -- here one, with itself -- and copies inside the device) or by direct copies.  host/multi_device.cpp is the user of these entry points (tests/test_gpu_dropin_binary.py runs it end to end); the
all-gather there replaces the reference's exchange of depth maps through depths.dmb files (APD.cpp:497-500)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lib(pkg):
    L = pkg.lib()
    L.apd_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.apd_device_free.argtypes = [C.c_int, C.c_void_p]
    L.apd_device_memcpy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.apd_device_memset.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t]
    L.apd_rescale_nearest_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.apd_exchange_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int]
    L.apd_exchange_allgather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]
    L.apd_exchange_backend.argtypes = [C.c_void_p]
    L.apd_exchange_backend.restype = C.c_char_p
    L.apd_exchange_destroy.argtypes = [C.c_void_p]
    L.apd_exchange_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.apd_exchange_last_error.restype = C.c_char_p
    return L


def _malloc(L, nbytes):
    p = C.c_void_p()
    assert L.apd_device_malloc(0, nbytes, C.byref(p)) == 0
    return p


def _host_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("devices,prefer_rccl,backend", [([0], 1, b"rccl"), ([0], 0, b"peer-copy"), ([0, 0], 1, b"rccl"),
                                                          ([0, 0, 0], 1, b"rccl"), ([0, 0, 0], 0, b"peer-copy"),
                                                          ([0] * 8, 1, b"rccl"), ([0] * 8, 0, b"peer-copy")])
def test_allgather_waits_for_the_copies_that_fill_its_send_buffers(gpu_pkg, devices, prefer_rccl, backend):
    """The regression behind the synchronisation in apd_exchange_allgather: the send buffers are packed with device-to-device
    copies on the null stream right before the exchange, whose own streams are non-blocking.  At 3100 x 2065 the planes of the
    later views used to arrive partly or not at all (a 1100 x 64 scene was too small to lose the race).  Large buffers,
    packed immediately before every exchange, several rounds."""
    L = _lib(gpu_pkg)
    n = len(devices)
    per_rank = (96 << 20) if n <= 3 else (24 << 20) + 16 * 7   # eight ranks: rank 7 exists before the driver's 8-GPU run does
    x = C.c_void_p()
    dev = (C.c_int * n)(*devices)
    assert L.apd_exchange_create(C.byref(x), n, dev, prefer_rccl) == 0, L.apd_exchange_last_error()
    assert L.apd_exchange_backend(x) == backend
    rng = np.random.default_rng(3)
    src = [_malloc(L, per_rank) for _ in range(n)]
    send = [_malloc(L, per_rank) for _ in range(n)]
    recv = [_malloc(L, per_rank * n) for _ in range(n)]
    try:
        for round_ in range(3):
            host = [rng.integers(0, 256, per_rank, dtype=np.uint8) for _ in range(n)]
            for r in range(n):
                assert L.apd_device_memcpy(0, src[r], _host_ptr(host[r]), per_rank) == 0
                assert L.apd_device_memset(0, recv[r], 0xEE, per_rank * n) == 0
            for r in range(n):   # the pack step of host/multi_device.cpp: device to device, then straight into the exchange
                assert L.apd_device_memset(0, send[r], 0, per_rank) == 0
                assert L.apd_device_memcpy(0, send[r], src[r], per_rank) == 0
            sp = (C.c_void_p * n)(*[s.value for s in send])
            rp = (C.c_void_p * n)(*[r_.value for r_ in recv])
            assert L.apd_exchange_allgather(x, sp, rp, per_rank) == 0, L.apd_exchange_last_error()
            want = np.concatenate(host)
            for r in range(n):
                got = np.empty(per_rank * n, np.uint8)
                assert L.apd_device_memcpy(0, _host_ptr(got), recv[r], per_rank * n) == 0
                assert np.array_equal(got, want), (round_, r, int((got != want).sum()))
        a, b = C.c_int(), C.c_int()
        assert L.apd_exchange_counts(x, C.byref(a), C.byref(b)) == 0
        assert (a.value, b.value) == ((3, 0) if backend == b"rccl" else (0, 3))
    finally:
        for p in src + send + recv:
            L.apd_device_free(0, p)
        L.apd_exchange_destroy(x)


def _rescale_reference(src, dw, dh):
    """RescaleMatToTargetSize (APD.cpp:752-774) with its swapped factors (row / scale_x, column / scale_y; pixels whose source
    index falls outside stay 0), as restated on the host in apd-mvs_amd/pipeline.py: the checker here."""
    import importlib
    pipeline = importlib.import_module("apd-mvs_amd.pipeline")
    return pipeline.rescale_nearest(src, dw, dh)


@pytest.mark.parametrize("elem,dtype,ch", [(1, np.uint8, 1), (4, np.uint32, 1), (16, np.float32, 4)])
def test_rescale_nearest_on_the_device_equals_the_host_restatement(gpu_pkg, elem, dtype, ch):
    L = _lib(gpu_pkg)
    rng = np.random.default_rng(9)
    for (sw, sh), (dw, dh) in (((155, 103), (310, 207)), ((388, 258), (775, 516)), ((64, 48), (64, 48)), ((775, 516), (1550, 1033))):
        shape = (sh, sw, ch) if ch > 1 else (sh, sw)
        a = rng.integers(0, 250, shape).astype(dtype)
        want = _rescale_reference(a, dw, dh)
        ds, dd = _malloc(L, a.nbytes), _malloc(L, want.nbytes)
        try:
            assert L.apd_device_memcpy(0, ds, _host_ptr(a), a.nbytes) == 0
            assert L.apd_rescale_nearest_device(0, ds, sw, sh, dd, dw, dh, elem) == 0
            got = np.empty_like(want)
            assert L.apd_device_memcpy(0, _host_ptr(got), dd, want.nbytes) == 0
            assert np.array_equal(got, want), ((sw, sh), (dw, dh), elem)
        finally:
            L.apd_device_free(0, ds)
            L.apd_device_free(0, dd)
    assert L.apd_rescale_nearest_device(0, None, 4, 4, None, 8, 8, 4) != 0
    p, q = _malloc(L, 64), _malloc(L, 64)
    assert L.apd_rescale_nearest_device(0, p, 2, 2, q, 4, 4, 3) != 0          # 3-byte elements are not a map type
    L.apd_device_free(0, p)
    L.apd_device_free(0, q)


def test_exchange_refuses_devices_that_do_not_exist(gpu_pkg):
    L = _lib(gpu_pkg)
    x = C.c_void_p()
    dev = (C.c_int * 2)(0, 99)
    assert L.apd_exchange_create(C.byref(x), 2, dev, 1) != 0
    assert b"99" in L.apd_exchange_last_error()


def test_allgather_after_waits_for_pending_writer_events(gpu_pkg, synth):
    """apd_exchange_allgather_after does no device-wide synchronisation: its streams wait for the events the caller names.  The send
    buffers are written on a side stream BEHIND a long queue of other work and the event is still pending when the exchange is
    called; the gathered bytes must be the ones written after it.  Both backends, two and eight ranks on the one device.  (ADVICE r05:
    the predecessor, apd_exchange_allgather_ready, relied on the caller having synchronised its streams.)"""
    import torch
    L = _lib(gpu_pkg)
    L.apd_exchange_allgather_after.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.apd_exchange_setup_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    dev = torch.device("cuda", 0)
    seen_pending = 0
    for n, prefer_rccl in ((2, 0), (8, 0), (2, 1), (8, 1)):
        per_rank = 24 << 20
        x = C.c_void_p()
        devs = (C.c_int * n)(*([0] * n))
        assert L.apd_exchange_create(C.byref(x), n, devs, prefer_rccl) == 0, L.apd_exchange_last_error()
        assert L.apd_exchange_backend(x) == (b"rccl" if prefer_rccl else b"peer-copy")
        dl, init = C.c_double(), C.c_double()
        assert L.apd_exchange_setup_times(x, C.byref(dl), C.byref(init)) == 0 and (init.value > 0) == bool(prefer_rccl)
        g = torch.Generator(device="cpu").manual_seed(17 + n)
        src = [torch.randint(0, 256, (per_rank,), dtype=torch.uint8, generator=g).to(dev) for _ in range(n)]
        send = [torch.zeros(per_rank, dtype=torch.uint8, device=dev) for _ in range(n)]
        recv = [torch.full((per_rank * n,), 0xEE, dtype=torch.uint8, device=dev) for _ in range(n)]
        burn = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        ev = torch.cuda.Event()
        try:
            for round_ in range(2):
                with torch.cuda.stream(side):
                    acc = burn
                    for _ in range(150):         # hundreds of milliseconds of queued work ahead of the writes
                        acc = acc @ burn
                        acc = acc / acc.abs().max()
                    for r in range(n):
                        send[r].copy_(src[r] if round_ == 0 else src[(r + 1) % n])
                    ev.record(side)
                pending = not ev.query()
                sp = (C.c_void_p * n)(*[t.data_ptr() for t in send])
                rp = (C.c_void_p * n)(*[t.data_ptr() for t in recv])
                evs = (C.c_void_p * 2)(None, ev.cuda_event)   # NULL entries are skipped
                assert L.apd_exchange_allgather_after(x, sp, rp, per_rank, 2, evs) == 0, L.apd_exchange_last_error()
                seen_pending += 1 if pending else 0
                want = torch.cat([src[r] if round_ == 0 else src[(r + 1) % n] for r in range(n)])
                for r in range(n):
                    assert torch.equal(recv[r], want), (n, prefer_rccl, round_, r)
            assert L.apd_exchange_allgather_after(x, sp, rp, per_rank, -1, None) != 0
        finally:
            torch.cuda.synchronize()
            assert L.apd_exchange_destroy(x) == 0
    # the wait must have been exercised: the writer's event was still pending when the exchange was called (every round but the ones
    # whose first matrix product pays a library's one-time set-up on the host; eight of eight on the builder's boxes)
    assert seen_pending >= 2, seen_pending


def test_export_event_marks_the_handles_last_export(gpu_pkg, synth):
    """apd_export_event: NULL before the first export, then the event recorded behind every export kernel -- what the scheduler hands
    to apd_exchange_allgather_after."""
    import torch
    import common
    L = gpu_pkg.lib()
    L.apd_export_event.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    W, H, N = 64, 48, 2
    sc, imgs = common.scene_inputs(synth, W, H, N)
    h = common.make_handle(gpu_pkg, sc, imgs, N, common.base_params(sc, N, max_iterations=1))
    ev = C.c_void_p(1)
    assert L.apd_export_event(h._h, C.byref(ev)) == 0 and ev.value is None
    h.run()
    d = torch.empty((H, W), device="cuda", dtype=torch.float32)
    h.export_depth_normal(d, None)
    assert L.apd_export_event(h._h, C.byref(ev)) == 0 and ev.value
    first = ev.value
    h.export_depth_normal(d, None)
    assert L.apd_export_event(h._h, C.byref(ev)) == 0 and ev.value == first   # one event per handle, re-recorded
    assert L.apd_export_event(None, C.byref(ev)) != 0
    h.close()
