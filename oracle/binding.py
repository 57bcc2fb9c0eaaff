"""ctypes binding of the CPU oracle (oracle/apd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libapd_oracle.so")

FIRST_INIT, REFINE_INIT, REFINE_ITER = 0, 1, 2
WEAK, STRONG, UNKNOWN = 0, 1, 2


class Camera(C.Structure):
    """Reference Camera (main.h:47-56), 112 bytes."""

    _fields_ = [
        ("K", C.c_float * 9),
        ("R", C.c_float * 9),
        ("t", C.c_float * 3),
        ("c", C.c_float * 3),
        ("height", C.c_int),
        ("width", C.c_int),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
    ]


class Params(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("num_images", C.c_int),
        ("top_k", C.c_int),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
        ("geom_consistency", C.c_int),
        ("strong_radius", C.c_int),
        ("strong_increment", C.c_int),
        ("weak_radius", C.c_int),
        ("weak_increment", C.c_int),
        ("use_APD", C.c_int),
        ("weak_peak_radius", C.c_int),
        ("rotate_time", C.c_int),
        ("ransac_threshold", C.c_float),
        ("geom_factor", C.c_float),
        ("state", C.c_int),
        ("seed", C.c_uint64),
    ]


def default_params(**kw):
    """Defaults of PatchMatchParams (main.h:75-94)."""
    p = Params(
        max_iterations=3, num_images=5, top_k=4, depth_min=0.0, depth_max=1.0, geom_consistency=0,
        strong_radius=5, strong_increment=2, weak_radius=5, weak_increment=5, use_APD=1,
        weak_peak_radius=2, rotate_time=4, ransac_threshold=0.005, geom_factor=0.2, state=FIRST_INIT,
        seed=12345,
    )
    for k, v in kw.items():
        setattr(p, k, v)
    return p


_FUSION_LIB_PATH = os.path.join(_HERE, "_build", "libapd_fusion_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("apd_oracle.c", "apd_oracle.h", "fusion_oracle.cpp", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH) and os.path.exists(_FUSION_LIB_PATH)
            and all(min(os.path.getmtime(_LIB_PATH), os.path.getmtime(_FUSION_LIB_PATH)) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fpp = C.POINTER(C.POINTER(C.c_float))
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.POINTER(Camera), fpp, fpp,
                             C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_get_threads.restype = C.c_int
    L.orc_set_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_set_study_weights_q8.argtypes = [C.c_int]
    L.orc_get_study_weights_q8.restype = C.c_int
    L.orc_run_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_run.argtypes = [C.c_void_p]
    L.orc_run_sweeps.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for name, rt in (("orc_planes", C.c_float), ("orc_fit_planes", C.c_float), ("orc_costs", C.c_float),
                     ("orc_rng", C.c_uint32), ("orc_selected_views", C.c_uint32),
                     ("orc_view_weight", C.c_uint8), ("orc_weak_info", C.c_uint8),
                     ("orc_weak_reliable", C.c_uint8), ("orc_nearest_strong", C.c_int16),
                     ("orc_neighbours_map", C.c_int32), ("orc_neighbours", C.c_int16)):
        f = getattr(L, name)
        f.restype = C.POINTER(rt)
        f.argtypes = [C.c_void_p]
    L.orc_weak_count.restype = C.c_int
    L.orc_weak_count.argtypes = [C.c_void_p]
    L.orc_xorwow_init.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    L.orc_xorwow_next.restype = C.c_uint32
    L.orc_xorwow_next.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_xorwow_uniform.restype = C.c_float
    L.orc_xorwow_uniform.argtypes = [C.POINTER(C.c_uint32)]
    for name in ("orc_sinf", "orc_cosf", "orc_expf"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    L.orc_homography.argtypes = [C.POINTER(Camera), C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_sample_bilinear.restype = C.c_float
    L.orc_sample_bilinear.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float]
    for name in ("orc_ncc_old", "orc_ncc_new", "orc_geom_cost"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.orc_depth_from_plane.restype = C.c_float
    L.orc_depth_from_plane.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.c_int, C.c_int]
    L.orc_distance_to_origin.restype = C.c_float
    L.orc_distance_to_origin.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    _lib = L
    return L


def make_camera(K, R, t, width, height, depth_min, depth_max):
    """Fills a Camera the way ReadCamera does (APD.cpp:51-92): c = -R^T t evaluated in double."""
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


def _fpp(arrays):
    arr = (C.POINTER(C.c_float) * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data_as(C.POINTER(C.c_float))
    return arr


class Oracle:
    """One (view, pass) of the reference's APD object, on the CPU oracle."""

    def __init__(self, width, height, params, cameras, images, depths=None, prior_planes=None,
                 prior_views=None, prior_weak=None):
        L = lib()
        self.W, self.H = int(width), int(height)
        self.params = params
        n = params.num_images
        assert len(cameras) == n and len(images) == n
        self._imgs = [np.ascontiguousarray(im, np.float32).reshape(self.H, self.W) for im in images]
        self._deps = None if depths is None else [np.ascontiguousarray(d, np.float32).reshape(self.H, self.W) for d in depths]
        cam_arr = (Camera * n)(*cameras)
        pp = None if prior_planes is None else np.ascontiguousarray(prior_planes, np.float32)
        pv = None if prior_views is None else np.ascontiguousarray(prior_views, np.uint32)
        pw = None if prior_weak is None else np.ascontiguousarray(prior_weak, np.uint8)
        self._h = L.orc_create(self.W, self.H, C.byref(params), cam_arr, _fpp(self._imgs),
                               None if self._deps is None else _fpp(self._deps),
                               None if pp is None else pp.ctypes.data,
                               None if pv is None else pv.ctypes.data,
                               None if pw is None else pw.ctypes.data)
        if not self._h:
            raise RuntimeError("orc_create failed")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_kernel(self, kid, it=0):
        lib().orc_run_kernel(self._h, kid, it)

    def set_roi(self, x0=0, y0=0, x1=None, y1=None):
        """Kernels visit only the pixels of [x0, x1) x [y0, y1) from now on (default: the whole image again)."""
        lib().orc_set_roi(self._h, x0, y0, self.W if x1 is None else x1, self.H if y1 is None else y1)

    def run(self):
        lib().orc_run(self._h)

    def run_sweeps(self, first_iter, iters):
        lib().orc_run_sweeps(self._h, first_iter, iters)

    def _view(self, fn, shape, dtype):
        ptr = getattr(lib(), fn)(self._h)
        return np.ctypeslib.as_array(ptr, shape=shape).view(dtype)

    @property
    def planes(self):
        return self._view("orc_planes", (self.H, self.W, 4), np.float32)

    @property
    def fit_planes(self):
        return self._view("orc_fit_planes", (self.H, self.W, 4), np.float32)

    @property
    def costs(self):
        return self._view("orc_costs", (self.H, self.W), np.float32)

    @property
    def rng(self):
        return self._view("orc_rng", (self.H, self.W, 6), np.uint32)

    @property
    def selected_views(self):
        return self._view("orc_selected_views", (self.H, self.W), np.uint32)

    @property
    def view_weight(self):
        return self._view("orc_view_weight", (self.H, self.W, 32), np.uint8)

    @property
    def weak_info(self):
        return self._view("orc_weak_info", (self.H, self.W), np.uint8)

    @property
    def weak_reliable(self):
        return self._view("orc_weak_reliable", (self.H, self.W), np.uint8)

    @property
    def nearest_strong(self):
        return self._view("orc_nearest_strong", (self.H, self.W, 2), np.int16)

    @property
    def neighbours_map(self):
        return self._view("orc_neighbours_map", (self.H, self.W), np.int32)

    @property
    def weak_count(self):
        return lib().orc_weak_count(self._h)

    @property
    def neighbours(self):
        n = max(self.weak_count, 1)
        return self._view("orc_neighbours", (n, 9, 2), np.int16)

    def ncc_old(self, x, y, src, plane):
        p = (C.c_float * 4)(*[float(v) for v in plane])
        return lib().orc_ncc_old(self._h, x, y, src, p)

    def ncc_new(self, x, y, src, plane):
        p = (C.c_float * 4)(*[float(v) for v in plane])
        return lib().orc_ncc_new(self._h, x, y, src, p)

    def geom_cost(self, x, y, src, plane):
        p = (C.c_float * 4)(*[float(v) for v in plane])
        return lib().orc_geom_cost(self._h, x, y, src, p)


def fuse(cameras, images, depths, normals, weaks, pairs, ply_path, blocks=None):
    """The reference's sequential fusion loop (oracle/fusion_oracle.cpp, RunFusion APD.cpp:826-977) on host arrays:
    cameras = ctypes array of Camera-compatible structs (one per view), images float32 [H, W] (grey) or [H, W, 3] (blue,
    green, red), depths float32 [H, W], normals float32
    [H, W, 3], weaks uint8 [H, W], pairs[i] = source view indices of view i.  Writes `ply_path`, returns the point count."""
    build()
    L = C.CDLL(_FUSION_LIB_PATH)
    L.orc_fuse.restype = C.c_longlong
    V = len(images)
    keep = []

    def ptrs(arrs, dt):
        out = (C.c_void_p * V)()
        for i, a in enumerate(arrs):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            out[i] = a.ctypes.data
        return out

    rows = (C.c_int * V)(*[d.shape[0] for d in depths])
    cols = (C.c_int * V)(*[d.shape[1] for d in depths])
    offs = (C.c_int * (V + 1))()
    flat = []
    for v in range(V):
        offs[v] = len(flat)
        flat += list(pairs[v])
    offs[V] = len(flat)
    idx = (C.c_int * max(len(flat), 1))(*flat)
    channels = 3 if np.asarray(images[0]).ndim == 3 else 1
    n = L.orc_fuse(V, C.byref(cameras), ptrs(images, np.float32), channels, ptrs(depths, np.float32), ptrs(normals, np.float32),
                   ptrs(weaks, np.uint8), None if blocks is None else ptrs(blocks, np.uint8), rows, cols, offs, idx,
                   str(ply_path).encode())
    if n < 0:
        raise IOError("cannot write " + str(ply_path))
    return int(n)
