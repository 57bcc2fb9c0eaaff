"""apd_mvs_amd -- thin ctypes layer over the C ABI (include/apd_mi355x.h) of the MI355X PatchMatch path.

This is plumbing for tests and bench.py only; the product is the HIP library
(`_build/libapd_mi355x.so`) and the C++ drop-in host in `host/`.  There is no CPU fallback: if the
library is missing or a GPU call fails, this module raises.

The directory is called `apd-mvs_amd` (not importable by name); `__graft_entry__.load_package()`
registers it as the module `apd_mvs_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libapd_mi355x.so")

FIRST_INIT, REFINE_INIT, REFINE_ITER = 0, 1, 2
WEAK, STRONG, UNKNOWN = 0, 1, 2
MAX_IMAGES = 32

(K1, K2, K3, K4, K5, K6, K7, K8, K9, K10, K11, K12, K13, K14, K15) = range(1, 16)
KERNEL_NAMES = {
    1: "InitRandomStates", 2: "FindNearestStrongPoint", 3: "GenNeighbours", 4: "NeigbourUpdate",
    5: "RandomInitialization", 6: "BlackPixelUpdateStrong", 7: "RedPixelUpdateStrong",
    8: "RANSACToGetFitPlane", 9: "BlackPixelUpdateWeak", 10: "RedPixelUpdateWeak",
    11: "GetDepthandNormal", 12: "BlackPixelFilterStrong", 13: "RedPixelFilterStrong",
    14: "DepthToWeak", 15: "LocalRefine",
}

# apd_set_option / apd_get_option (include/apd_mi355x.h): the library reads nothing from the environment
(OPT_FAST_RCP, OPT_EARLY_OUT, OPT_SOURCE_QUADS, OPT_TILED_COPY, OPT_K67_WINDOWS, OPT_K1415_WINDOWS) = range(6)
OPTION_NAMES = {"fast_rcp": OPT_FAST_RCP, "early_out": OPT_EARLY_OUT, "source_quads": OPT_SOURCE_QUADS, "tiled_copy": OPT_TILED_COPY,
                "k67_windows": OPT_K67_WINDOWS, "k1415_windows": OPT_K1415_WINDOWS}

(STATE_PLANES, STATE_FIT_PLANES, STATE_COSTS, STATE_RNG, STATE_SELECTED_VIEWS, STATE_VIEW_WEIGHT,
 STATE_WEAK_INFO, STATE_WEAK_RELIABLE, STATE_NEAREST_STRONG, STATE_NEIGHBOURS_MAP, STATE_NEIGHBOURS) = range(11)


class Camera(C.Structure):
    """apd_camera == reference Camera (main.h:47-56)."""

    _fields_ = [("K", C.c_float * 9), ("R", C.c_float * 9), ("t", C.c_float * 3), ("c", C.c_float * 3),
                ("height", C.c_int), ("width", C.c_int), ("depth_min", C.c_float), ("depth_max", C.c_float)]


class Params(C.Structure):
    """apd_params == reference PatchMatchParams (main.h:75-94) + seed."""

    _fields_ = [
        ("max_iterations", C.c_int), ("num_images", C.c_int), ("sigma_spatial", C.c_float),
        ("sigma_color", C.c_float), ("top_k", C.c_int), ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("geom_consistency", C.c_int), ("strong_radius", C.c_int), ("strong_increment", C.c_int),
        ("weak_radius", C.c_int), ("weak_increment", C.c_int), ("use_APD", C.c_int),
        ("weak_peak_radius", C.c_int), ("rotate_time", C.c_int), ("ransac_threshold", C.c_float),
        ("geom_factor", C.c_float), ("state", C.c_int), ("seed", C.c_uint64),
    ]


class ApdError(RuntimeError):
    pass


_lib = None


def library_path():
    return LIB_PATH


def expected_build_id():
    """Digest of csrc/*, the headers and the compiler flags of THIS tree (build.py: expected_build_id)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("apd_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.expected_build_id()


def build_id(L=None):
    """apd_build_id() of the loaded library: the digest of the sources it was compiled from."""
    L = L or lib()
    L.apd_build_id.restype = C.c_char_p
    return L.apd_build_id().decode()


def check_build_id(L):
    """A library built from other sources than the tree's is refused (an mtime test cannot tell a stale binary on a box
    whose push preserved the times).  APD_ALLOW_STALE_LIBRARY=1 skips the test (lab builds with ad-hoc flags)."""
    if os.environ.get("APD_ALLOW_STALE_LIBRARY") == "1":
        return
    if not hasattr(L, "apd_build_id"):
        raise ApdError("%s has no apd_build_id(): a binary of an earlier round; run __graft_entry__.build()" % LIB_PATH)
    have, want = build_id(L), expected_build_id()
    if have != want:
        raise ApdError("stale HIP library: %s was built from sources with digest %s, the tree's csrc/ + flags give %s; "
                       "run __graft_entry__.build() (a lab build with APD_EXTRA_FLAGS: set APD_ALLOW_STALE_LIBRARY=1)" % (LIB_PATH, have, want))


def lib():
    """Loads the HIP library; raises if it has not been built or was built from other sources (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApdError("HIP library %s is missing: run __graft_entry__.build() first" % LIB_PATH)
    try:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; if it is going to
        # be used (streams, device tensors, RCCL) it must be the copy that gets loaded, so import it first.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    check_build_id(L)
    H = C.c_void_p
    fpp = C.POINTER(C.c_void_p)
    L.apd_default_params.argtypes = [C.POINTER(Params)]
    L.apd_default_params.restype = None
    L.apd_create.argtypes = [C.POINTER(H), C.c_int, C.c_int, C.c_int, C.POINTER(Params)]
    L.apd_destroy.argtypes = [H]
    L.apd_reset.argtypes = [H, C.POINTER(Params)]
    L.apd_upload_views.argtypes = [H, C.c_int, C.POINTER(Camera), fpp, fpp]
    L.apd_upload_views_split.argtypes = [H, C.c_int, C.POINTER(Camera), fpp]
    L.apd_image_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.apd_image_destroy.argtypes = [C.c_void_p]
    L.apd_upload_views_shared.argtypes = [H, C.c_int, C.POINTER(Camera), fpp]
    L.apd_upload_depths.argtypes = [H, C.c_int, fpp]
    L.apd_run_before_depths.argtypes = [H]
    L.apd_run_after_depths.argtypes = [H]
    L.apd_upload_prior.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.apd_run.argtypes = [H]
    L.apd_run_kernel.argtypes = [H, C.c_int, C.c_int]
    L.apd_run_sweeps.argtypes = [H, C.c_int, C.c_int]
    L.apd_synchronize.argtypes = [H]
    L.apd_download.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.apd_download_state.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    L.apd_upload_state.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    L.apd_state_bytes.argtypes = [H, C.c_int]
    L.apd_state_bytes.restype = C.c_size_t
    L.apd_export_depth_normal_device.argtypes = [H, C.c_void_p, C.c_void_p]
    for n in ("apd_width", "apd_height", "apd_weak_count"):
        getattr(L, n).argtypes = [H]
    for n in ("apd_depth_min", "apd_depth_max"):
        getattr(L, n).argtypes = [H]
        getattr(L, n).restype = C.c_float
    L.apd_profile_enable.argtypes = [H, C.c_int]
    L.apd_profile_reset.argtypes = [H]
    L.apd_profile_get.argtypes = [H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.apd_set_stream.argtypes = [H, C.c_void_p]
    L.apd_set_option.argtypes = [H, C.c_int, C.c_int]
    L.apd_get_option.argtypes = [H, C.c_int, C.POINTER(C.c_int)]
    L.apd_last_error.restype = C.c_char_p
    L.apd_device_count.restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise ApdError("apd error %d: %s" % (rc, lib().apd_last_error().decode()))


def default_params(**kw):
    p = Params()
    lib().apd_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_camera(K, R, t, width, height, depth_min, depth_max):
    """Fills a camera the way ReadCamera does (APD.cpp:51-92): c = -R^T t evaluated in double."""
    cam = Camera()
    K = np.asarray(K, np.float32).reshape(9)
    R = np.asarray(R, np.float32).reshape(9)
    t = np.asarray(t, np.float32).reshape(3)
    for i in range(9):
        cam.K[i] = float(K[i])
        cam.R[i] = float(R[i])
    for i in range(3):
        cam.t[i] = float(t[i])
    Rd, td = R.astype(np.float64), t.astype(np.float64)
    for j in range(3):
        cam.c[j] = float(np.float32(-(Rd[0 + j] * td[0] + Rd[3 + j] * td[1] + Rd[6 + j] * td[2])))
    cam.width, cam.height = int(width), int(height)
    cam.depth_min, cam.depth_max = float(depth_min), float(depth_max)
    return cam


def _ptr(a):
    """Raw address of a numpy array or torch tensor (host or device), or None."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


_STATE_DTYPES = {
    STATE_PLANES: (np.float32, 4), STATE_FIT_PLANES: (np.float32, 4), STATE_COSTS: (np.float32, 1),
    STATE_RNG: (np.uint32, 6), STATE_SELECTED_VIEWS: (np.uint32, 1), STATE_VIEW_WEIGHT: (np.uint8, 32),
    STATE_WEAK_INFO: (np.uint8, 1), STATE_WEAK_RELIABLE: (np.uint8, 1), STATE_NEAREST_STRONG: (np.int16, 2),
    STATE_NEIGHBOURS_MAP: (np.int32, 1),
}


class SharedImage:
    """apd_image_create: one W x H float image (numpy or torch, host or device) on the device, tested and packed once."""

    def __init__(self, width, height, pixels, device=0):
        if hasattr(pixels, "data_ptr"):
            import torch
            pixels = pixels.to(torch.float32).contiguous()
            assert pixels.numel() == width * height
        else:
            pixels = np.ascontiguousarray(pixels, np.float32)
            assert pixels.size == width * height
        self._keep = pixels
        p = C.c_void_p()
        _check(lib().apd_image_create(C.byref(p), device, width, height, _ptr(pixels)))
        self.ptr = p.value

    def close(self):
        if self.ptr:
            lib().apd_image_destroy(self.ptr)
            self.ptr = None


class Handle:
    """One (reference view, pass): the C-ABI equivalent of the reference's `APD` object (APD.h:67-145)."""

    def __init__(self, width, height, params, device=-1):
        self._h = C.c_void_p()
        self.W, self.H = int(width), int(height)
        self.params = params
        self.device = int(device)
        self._keep = []
        _check(lib().apd_create(C.byref(self._h), device, self.W, self.H, C.byref(params)))

    def close(self):
        if self._h:
            lib().apd_destroy(self._h)
            self._h = C.c_void_p()

    def reset(self, params):
        """Re-arms the handle for another (view, pass) of the same size (apd_reset): same initial state as a new one."""
        self.params = params
        self._keep = []
        _check(lib().apd_reset(self._h, C.byref(params)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        """apd_set_option; `option` is an OPT_* constant or one of OPTION_NAMES.  Upload-time options (source_quads,
        tiled_copy) must be set before upload_views."""
        opt = OPTION_NAMES[option] if isinstance(option, str) else int(option)
        _check(lib().apd_set_option(self._h, opt, int(value)))

    def get_option(self, option):
        opt = OPTION_NAMES[option] if isinstance(option, str) else int(option)
        v = C.c_int()
        _check(lib().apd_get_option(self._h, opt, C.byref(v)))
        return v.value

    def upload_views(self, cameras, images, depths=None):
        """images/depths: lists of float32 [H, W] numpy arrays or torch tensors (host or device)."""
        n = len(cameras)
        imgs = [self._as_f32(im) for im in images]
        deps = None if depths is None else [self._as_f32(d) for d in depths]
        self._keep = [imgs, deps]
        cam_arr = (Camera * n)(*cameras)
        ip = (C.c_void_p * n)(*[_ptr(a) for a in imgs])
        dp = None if deps is None else (C.c_void_p * n)(*[_ptr(a) for a in deps])
        _check(lib().apd_upload_views(self._h, n, cam_arr, ip, dp))
        self.params.num_images = n

    def upload_views_split(self, cameras, images):
        """apd_upload_views_split: a geometric pass whose depth maps follow with upload_depths."""
        n = len(cameras)
        imgs = [self._as_f32(im) for im in images]
        self._keep = [imgs, None]
        cam_arr = (Camera * n)(*cameras)
        ip = (C.c_void_p * n)(*[_ptr(a) for a in imgs])
        _check(lib().apd_upload_views_split(self._h, n, cam_arr, ip))
        self.params.num_images = n

    def upload_views_shared(self, cameras, images):
        """apd_upload_views_shared: `images` are SharedImage objects (created once, used by any number of handles); in a geometric
        pass the depth maps follow with upload_depths."""
        n = len(cameras)
        self._keep = [list(images), None]
        cam_arr = (Camera * n)(*cameras)
        ip = (C.c_void_p * n)(*[im.ptr for im in images])
        _check(lib().apd_upload_views_shared(self._h, n, cam_arr, ip))
        self.params.num_images = n

    def upload_depths(self, depths):
        deps = [self._as_f32(d) for d in depths]
        dp = (C.c_void_p * len(deps))(*[_ptr(a) for a in deps])
        _check(lib().apd_upload_depths(self._h, len(deps), dp))

    def run_before_depths(self):
        _check(lib().apd_run_before_depths(self._h))

    def run_after_depths(self):
        _check(lib().apd_run_after_depths(self._h))
        self.synchronize()

    def _as_f32(self, a):
        if hasattr(a, "data_ptr"):
            import torch
            a = a.to(torch.float32).contiguous()
            assert a.numel() == self.W * self.H
            return a
        a = np.ascontiguousarray(a, np.float32)
        assert a.size == self.W * self.H
        return a

    def upload_prior(self, planes=None, selected_views=None, weak_info=None):
        """numpy arrays or torch tensors (host or device): planes float32 [H, W, 4], selected views 32-bit, weak uint8."""
        def prep(a, np_dtype, itemsize):
            if a is None:
                return None
            if hasattr(a, "data_ptr"):
                a = a.contiguous()
                assert a.element_size() == itemsize
                return a
            return np.ascontiguousarray(a, np_dtype)
        p, v, w = prep(planes, np.float32, 4), prep(selected_views, np.uint32, 4), prep(weak_info, np.uint8, 1)
        self._keep_prior = (p, v, w)
        _check(lib().apd_upload_prior(self._h, _ptr(p), _ptr(v), _ptr(w)))

    def run(self):
        _check(lib().apd_run(self._h))
        self.synchronize()

    def run_kernel(self, kid, it=0, sync=True):
        _check(lib().apd_run_kernel(self._h, kid, it))
        if sync:
            self.synchronize()

    def run_sweeps(self, first_iter, iters, sync=True):
        _check(lib().apd_run_sweeps(self._h, first_iter, iters))
        if sync:
            self.synchronize()

    def synchronize(self):
        _check(lib().apd_synchronize(self._h))

    def download(self):
        n = self.W * self.H
        planes = np.empty((self.H, self.W, 4), np.float32)
        weak = np.empty((self.H, self.W), np.uint8)
        views = np.empty((self.H, self.W), np.uint32)
        _check(lib().apd_download(self._h, planes.ctypes.data, weak.ctypes.data, views.ctypes.data))
        return planes, weak, views

    def download_device(self):
        """The same three maps as torch tensors on the handle's device (device-to-device copies): planes float32
        [H, W, 4], weak uint8 [H, W], selected views as int32 bit patterns [H, W]."""
        import torch
        dev = torch.device("cuda", self.device if self.device >= 0 else torch.cuda.current_device())
        planes = torch.empty((self.H, self.W, 4), dtype=torch.float32, device=dev)
        weak = torch.empty((self.H, self.W), dtype=torch.uint8, device=dev)
        views = torch.empty((self.H, self.W), dtype=torch.int32, device=dev)
        _check(lib().apd_download(self._h, planes.data_ptr(), weak.data_ptr(), views.data_ptr()))
        return planes, weak, views

    def state(self, which):
        nbytes = lib().apd_state_bytes(self._h, which)
        if which == STATE_NEIGHBOURS:
            out = np.empty((nbytes // 36, 9, 2), np.int16)
        else:
            dt, k = _STATE_DTYPES[which]
            shape = (self.H, self.W, k) if k > 1 else (self.H, self.W)
            out = np.empty(shape, dt)
        assert out.nbytes == nbytes, (which, out.nbytes, nbytes)
        _check(lib().apd_download_state(self._h, which, out.ctypes.data, nbytes))
        return out

    def set_state(self, which, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().apd_upload_state(self._h, which, arr.ctypes.data, arr.nbytes))

    def export_depth_normal(self, depth_dev, normal_dev=None):
        """depth_dev / normal_dev: torch CUDA tensors ([H,W] and [H,W,3] float32)."""
        _check(lib().apd_export_depth_normal_device(self._h, _ptr(depth_dev), _ptr(normal_dev)))

    @property
    def weak_count(self):
        return lib().apd_weak_count(self._h)

    def profile_enable(self, on=True):
        _check(lib().apd_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        _check(lib().apd_profile_reset(self._h))

    def profile(self):
        """{kernel_id: (total_ms, launches)} accumulated since the last reset."""
        out = {}
        for k in range(1, 16):
            ms, n = C.c_double(), C.c_int()
            _check(lib().apd_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            if n.value:
                out[k] = (ms.value, n.value)
        return out


def device_count():
    return lib().apd_device_count()
