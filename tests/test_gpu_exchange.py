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
                                                          ([0, 0, 0], 1, b"rccl"), ([0, 0, 0], 0, b"peer-copy")])
def test_allgather_waits_for_the_copies_that_fill_its_send_buffers(gpu_pkg, devices, prefer_rccl, backend):
    """The regression behind the synchronisation in apd_exchange_allgather: the send buffers are packed with device-to-device
    copies on the null stream right before the exchange, whose own streams are non-blocking.  At 3100 x 2065 the planes of the
    later views used to arrive partly or not at all (a 1100 x 64 scene was too small to lose the race).  Large buffers,
    packed immediately before every exchange, several rounds."""
    L = _lib(gpu_pkg)
    n = len(devices)
    per_rank = 96 << 20
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


def test_async_set_up_serves_exchanges_with_copies_until_rccl_is_ready(gpu_pkg):
    """apd_exchange_create_async returns at once; every all-gather gives the same bytes whether it ran through direct copies (RCCL
    still initialising) or through RCCL (ready); apd_exchange_wait ends the set-up and reports what it took; after it the backend is
    RCCL.  apd_exchange_allgather_ready (no device-wide synchronisation: the caller has synchronised the writers) gives the same bytes."""
    L = _lib(gpu_pkg)
    L.apd_exchange_create_async.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int]
    L.apd_exchange_wait.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.apd_exchange_setup_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.apd_exchange_allgather_ready.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]
    assert L.apd_exchange_preload_rccl() == 0
    n, per_rank = 2, 8 << 20
    x = C.c_void_p()
    dev = (C.c_int * n)(0, 0)
    assert L.apd_exchange_create_async(C.byref(x), n, dev, 1) == 0, L.apd_exchange_last_error()
    rng = np.random.default_rng(11)
    send = [_malloc(L, per_rank) for _ in range(n)]
    recv = [_malloc(L, per_rank * n) for _ in range(n)]
    try:
        sp = (C.c_void_p * n)(*[s.value for s in send])
        rp = (C.c_void_p * n)(*[r_.value for r_ in recv])

        def one_round(fn):
            host = [rng.integers(0, 256, per_rank, dtype=np.uint8) for _ in range(n)]
            for r in range(n):
                assert L.apd_device_memcpy(0, send[r], _host_ptr(host[r]), per_rank) == 0   # synchronises: the sends are complete
                assert L.apd_device_memset(0, recv[r], 0xEE, per_rank * n) == 0
            assert fn(x, sp, rp, per_rank) == 0, L.apd_exchange_last_error()
            want = np.concatenate(host)
            for r in range(n):
                got = np.empty(per_rank * n, np.uint8)
                assert L.apd_device_memcpy(0, _host_ptr(got), recv[r], per_rank * n) == 0
                assert np.array_equal(got, want)

        one_round(L.apd_exchange_allgather)          # usually while RCCL is still initialising: direct copies
        one_round(L.apd_exchange_allgather_ready)
        setup, waited = C.c_double(), C.c_double()
        assert L.apd_exchange_wait(x, C.byref(setup), C.byref(waited)) == 0
        assert setup.value > 0 and waited.value >= 0
        assert L.apd_exchange_backend(x) == b"rccl"
        dl, init = C.c_double(), C.c_double()
        assert L.apd_exchange_setup_times(x, C.byref(dl), C.byref(init)) == 0 and dl.value > 0 and init.value > 0
        one_round(L.apd_exchange_allgather)          # RCCL now
        one_round(L.apd_exchange_allgather_ready)
        a, b = C.c_int(), C.c_int()
        assert L.apd_exchange_counts(x, C.byref(a), C.byref(b)) == 0
        assert a.value + b.value == 4 and a.value >= 2
    finally:
        for p in send + recv:
            L.apd_device_free(0, p)
        assert L.apd_exchange_destroy(x) == 0
