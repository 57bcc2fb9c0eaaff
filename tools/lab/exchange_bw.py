"""Time apd_exchange_allgather (device-wide form) on an idle device: ranks x bytes per rank, both backends.  Usage: python tools/lab/exchange_bw.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
L = pkg.lib()
L.apd_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
L.apd_exchange_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int]
L.apd_exchange_allgather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]
L.apd_exchange_allgather_after.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
L.apd_exchange_destroy.argtypes = [C.c_void_p]
L.apd_device_free.argtypes = [C.c_int, C.c_void_p]
for n, per_rank in ((2, 100 << 20), (8, 25 << 20), (2, 8 << 20)):
    for rccl in (0, 1):
        x = C.c_void_p()
        assert L.apd_exchange_create(C.byref(x), n, (C.c_int * n)(*([0] * n)), rccl) == 0
        send, recv = [], []
        for _ in range(n):
            p = C.c_void_p(); assert L.apd_device_malloc(0, per_rank, C.byref(p)) == 0; send.append(p)
            q = C.c_void_p(); assert L.apd_device_malloc(0, per_rank * n, C.byref(q)) == 0; recv.append(q)
        sp = (C.c_void_p * n)(*[s.value for s in send]); rp = (C.c_void_p * n)(*[r.value for r in recv])
        for fn, name in ((lambda: L.apd_exchange_allgather(x, sp, rp, per_rank), "device-sync"), (lambda: L.apd_exchange_allgather_after(x, sp, rp, per_rank, 0, None), "after(events)")):
            fn()
            t0 = time.perf_counter()
            for _ in range(10):
                assert fn() == 0
            dt = (time.perf_counter() - t0) / 10
            print("%d ranks x %3d MB, %-9s %-13s %7.2f ms per all-gather (%.0f GB/s written)" % (n, per_rank >> 20, "rccl" if rccl else "peer-copy", name, dt * 1e3, n * n * per_rank / dt / 1e9))
        for p in send + recv:
            L.apd_device_free(0, p)
        L.apd_exchange_destroy(x)
