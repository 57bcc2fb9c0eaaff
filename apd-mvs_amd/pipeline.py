"""In-memory multi-view, multi-scale pass scheduler for the PatchMatch path, sharded over ranks.

Mirrors the reference's driver (main.cpp:140-217): `round_num` pyramid levels x (1 photometric + 3 geometric)
passes over every reference view, with the per-pass parameters of main.cpp:171-212 and the level handling of
APD::InuputInitialization (APD.cpp:464-581).  Two things differ from the reference, both by design:

* state between passes (depth, normal, weak map, selected views per view) stays in memory instead of travelling
  through depths.dmb / normals.dmb / weak.bin / selected_views.bin; decoded and rescaled images are cached per level
  instead of being re-read for every (view, pass);
* views are sharded round-robin over the ranks of a torch.distributed group (one process per GPU).  After every
  pass the ranks all-gather the depth maps (the geometric term of the next pass reads the sources' depth maps,
  APD.cpp:492-509, APD.cu:760-772) and, after the last pass, depth + normal + weak maps (what fusion consumes).

With one rank the order of evaluation is the reference's (Gauss-Seidel over views inside a pass: later views read the
depth maps earlier views have just written) and the results are bit-identical to the drop-in binary
(tests/test_gpu_dropin_binary.py).  With G ranks a view sees this pass's maps of the views its own rank has already
processed and the previous pass's maps of everybody else's.

The compute backend is injected: `HipBackend` (the product) drives the C ABI; the CPU tests plug the oracle in.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import sharding


@dataclass
class MvsScene:
    """cameras[i] (full resolution, `apd_camera`), images[i] (float32 [H, W]), pairs[i] = source view ids of view i.
    Views 0 .. num_views-1 are reference views (one pair.txt entry each).  Further cameras / images, if any, are SOURCE-ONLY
    views: images that pair.txt lists as sources but that have no entry of their own (a subset run).  The reference simply
    loads them (APD.cpp:419-452); they are never processed, contribute no depth map to the geometric term (the reference
    reads a file that does not exist there, APD.cpp:497-500) and take no part in the fusion."""
    cameras: list
    images: list
    pairs: list

    @property
    def num_views(self):
        return len(self.pairs)


@dataclass
class PassSpec:
    """One pass over all views (main.cpp:169-215)."""
    round_index: int
    iteration_index: int
    scale_size: int
    params: dict = field(default_factory=dict)


@dataclass
class ViewState:
    depth: np.ndarray      # float32 [H, W], 0 = invalid (main.cpp:109-112)
    normal: np.ndarray     # float32 [H, W, 3], world frame
    weak: np.ndarray       # uint8 [H, W]
    views: np.ndarray      # uint32 [H, W] selected-view bitmask


def compute_round_num(width, height):
    """main.cpp:72-88: halve until max(W, H) <= 1000 (host/schedule.h: RoundNum)."""
    return int(host_lib().apdhost_round_num(int(width), int(height)))


def pass_schedule(round_num, iters=3, single_level=False):
    """Per-pass parameters of main.cpp:168-215.  ONE implementation: the table host/schedule.h builds for the C++ schedulers
    (BuildSchedule), read through libapd_host.so; level 0 leaves ransac_threshold / rotate_time at the struct defaults, as the
    reference's long-lived PatchMatchParams does (schedule.h: Configure)."""
    L = host_lib()
    n = L.apdhost_schedule(int(round_num), 1 if single_level else 0, None, 0)
    rows = (C.c_int * (9 * n))()
    L.apdhost_schedule(int(round_num), 1 if single_level else 0, rows, n)
    out = []
    for k in range(n):
        level, iteration, scale, state, geom, use_apd, peak, rotate, thr_bits = rows[9 * k:9 * k + 9]
        p = dict(state=state, geom_consistency=geom, max_iterations=iters, weak_peak_radius=peak, use_APD=use_apd)
        if use_apd:
            p.update(ransac_threshold=float(np.array([thr_bits], np.int32).view(np.float32)[0]), rotate_time=rotate)
        out.append(PassSpec(level, iteration, scale, p))
    return out


def resize_linear(src, new_cols, new_rows):
    """cv::resize(..., INTER_LINEAR) on a float image as the C++ host does it (host/APD.cpp ResizeLinear):
    fx = (dx + 0.5) * (src/dst) - 0.5 in double, taps floor(fx), floor(fx)+1 clamped, float weights, rows after columns."""
    src = np.ascontiguousarray(src, np.float32)
    rows, cols = src.shape

    def taps(n_new, n_old):
        f = ((np.arange(n_new, dtype=np.float64) + 0.5) * (float(n_old) / n_new) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        a = (f - i.astype(np.float32)).astype(np.float32)
        lo = i < 0
        i[lo], a[lo] = 0, 0
        hi = i >= n_old - 1
        i[hi], a[hi] = n_old - 1, 0
        return i, np.minimum(i + 1, n_old - 1), a

    x0, x1, ax = taps(new_cols, cols)
    y0, y1, ay = taps(new_rows, rows)
    one = np.float32(1.0)
    r0, r1 = src[y0], src[y1]
    top = r0[:, x0] * (one - ax) + r0[:, x1] * ax
    bot = r1[:, x0] * (one - ax) + r1[:, x1] * ax
    ay = ay[:, None]
    return (top * (one - ay) + bot * ay).astype(np.float32)


def rescale_nearest(src, target_width, target_height):
    """RescaleMatToTargetSize (APD.cpp:752-774) including its swapped factors: row / scale_x, column / scale_y."""
    rows, cols = src.shape[:2]
    if cols == target_width and rows == target_height:
        return src
    if hasattr(src, "data_ptr"):  # torch tensor (host or device): the same index arithmetic in float32
        import torch
        sx = np.float32(target_width) / np.float32(cols)
        sy = np.float32(target_height) / np.float32(rows)
        o_r = (torch.arange(target_height, dtype=torch.float32) / float(sx)).to(torch.int64)
        o_c = (torch.arange(target_width, dtype=torch.float32) / float(sy)).to(torch.int64)
        ok_r, ok_c = o_r < rows, o_c < cols
        out = torch.zeros((target_height, target_width) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
        rr = o_r.clamp(max=rows - 1).to(src.device)
        cc = o_c.clamp(max=cols - 1).to(src.device)
        picked = src[rr][:, cc]
        mask = (ok_r[:, None] & ok_c[None, :]).to(src.device)
        out[mask] = picked[mask]
        return out
    scale_x = np.float32(target_width) / np.float32(cols)
    scale_y = np.float32(target_height) / np.float32(rows)
    o_r = (np.arange(target_height, dtype=np.float32) / scale_x).astype(np.int64)
    o_c = (np.arange(target_width, dtype=np.float32) / scale_y).astype(np.int64)
    out = np.zeros((target_height, target_width) + src.shape[2:], src.dtype)
    ok_r, ok_c = o_r < rows, o_c < cols
    out[np.ix_(ok_r, ok_c)] = src[np.ix_(o_r[ok_r], o_c[ok_c])]
    return out


def level_inputs(scene, scale_size, camera_type):
    """Scaled images and intrinsics of one pyramid level (APD.cpp:464-488): every image by its own rounded size."""
    cams, imgs = [], []
    for cam, img in zip(scene.cameras, scene.images):
        c = camera_type.from_buffer_copy(cam)
        rows, cols = img.shape
        c.width, c.height = cols, rows
        if scale_size != 1:
            factor = np.float32(1.0) / np.float32(scale_size)
            new_cols = int(math.floor(float(np.float32(cols) * factor) + 0.5))   # std::round of a positive float
            new_rows = int(math.floor(float(np.float32(rows) * factor) + 0.5))
            sx = np.float32(new_cols) / np.float32(cols)
            sy = np.float32(new_rows) / np.float32(rows)
            img = resize_linear(img, new_cols, new_rows)
            c.K[0] = float(np.float32(c.K[0]) * sx)
            c.K[2] = float(np.float32(c.K[2]) * sx)
            c.K[4] = float(np.float32(c.K[4]) * sy)
            c.K[5] = float(np.float32(c.K[5]) * sy)
            c.width, c.height = new_cols, new_rows
        cams.append(c)
        imgs.append(np.ascontiguousarray(img, np.float32))
    return cams, imgs


class HipBackend:
    """One (view, pass) on the MI355X through the C ABI (== ProcessProblem, main.cpp:91-115)."""

    accepts_tensors = True  # run_pass takes and returns torch tensors on `device`: no host copies between the passes

    def __init__(self, pkg, device=0):
        self.pkg = pkg
        self.device = device
        self._pool = {}  # (width, height) -> recycled handle: no hipMalloc / hipFree per (view, pass)

    def close(self):
        for h in self._pool.values():
            h.close()
        self._pool.clear()

    @property
    def camera_type(self):
        return self.pkg.Camera

    def run_pass(self, width, height, params, cameras, images, depths, prior):
        import torch
        # the inputs were produced on torch's stream, the handle copies and computes on its own: hand over on the host
        torch.cuda.current_stream(torch.device("cuda", self.device)).synchronize()
        pkg = self.pkg
        h = self._pool.get((width, height))
        if h is None:
            if len(self._pool) >= 2:  # one level at a time (+ views of another size): drop what is no longer used
                self.close()
            h = pkg.Handle(width, height, pkg.default_params(**params), device=self.device)
            self._pool[(width, height)] = h
        else:
            h.reset(pkg.default_params(**params))
        h.upload_views(cameras, images, depths)
        if prior is not None:
            h.upload_prior(*prior)
        h.run()
        return h.download_device()


def run_pipeline(scene, backend, iters=3, seed=12345, single_level=False, group=None, max_rounds=None, max_passes=None, log=None):
    """Runs every pass of the schedule on this rank's views; returns {view: ViewState} for ALL views on every rank."""
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    V = scene.num_views
    mine = sharding.shard_views(V, world, rank)
    h0, w0 = scene.images[0].shape
    round_num = 1 if single_level else compute_round_num(w0, h0)
    if max_rounds is not None:
        round_num = min(round_num, max_rounds)
    schedule = pass_schedule(round_num, iters, single_level)
    if max_passes is not None:
        schedule = schedule[:max_passes]
    # Everything between the passes lives in torch tensors on the backend's device (the GPU of this rank for the HIP
    # backend): level images, depth maps of all views, prior state, post-processing, nearest-neighbour upsampling
    # (APD.cpp:752-774) and the all-gathers.  Only the final result is copied to the host.
    on_gpu = getattr(backend, "device", None) is not None and torch.cuda.is_available()
    device = torch.device("cuda", backend.device) if on_gpu else torch.device("cpu")
    tensors_in = bool(getattr(backend, "accepts_tensors", False))

    def to_dev(a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        if dtype is not None:
            t = t.to(dtype)
        return t.to(device)

    def call_backend(W, H, p, cams_, imgs_, depths_, prior_):
        if tensors_in:
            return backend.run_pass(W, H, p, cams_, imgs_, depths_, prior_)
        # numpy backend (the CPU oracle in the tests): zero-copy views of the CPU tensors
        npy = lambda t: None if t is None else t.contiguous().numpy()
        pr = None if prior_ is None else (npy(prior_[0]), None if prior_[1] is None else npy(prior_[1]).view(np.uint32), npy(prior_[2]))
        planes, weak, views = backend.run_pass(W, H, p, cams_, [npy(t) for t in imgs_],
                                               None if depths_ is None else [npy(t) for t in depths_], pr)
        return to_dev(planes), to_dev(weak), to_dev(views.view(np.int32))

    state = {}        # own views: (planes4 = world normal xyz + depth w, weak, views as int32 bits) at the level last written
    depth_store = {}  # every view's depth map as this rank knows it
    level_cache = {}
    for spec in schedule:
        if spec.scale_size not in level_cache:
            level_cache.clear()
            cams, imgs = level_inputs(scene, spec.scale_size, backend.camera_type)
            level_cache[spec.scale_size] = (cams, [to_dev(im) for im in imgs])
        cams, imgs = level_cache[spec.scale_size]
        for idx in mine:
            order = [idx] + list(scene.pairs[idx])
            W, H = cams[idx].width, cams[idx].height
            p = dict(spec.params)
            p["num_images"] = len(order)
            p["depth_min"] = float(np.float32(cams[idx].depth_min) * np.float32(0.6))   # APD.cpp:454-455
            p["depth_max"] = float(np.float32(cams[idx].depth_max) * np.float32(1.2))
            p["seed"] = seed + spec.iteration_index * 7919 + idx
            depths = None
            if p["geom_consistency"]:
                for j in order:
                    if j >= V and j not in depth_store:  # source-only view: no estimate anywhere
                        depth_store[j] = torch.zeros((H, W), dtype=torch.float32, device=device)
                depths = [rescale_nearest(depth_store[j], W, H).contiguous() for j in order]
            prior = None
            if p["state"] != 0:
                planes0, weak0, views0 = state[idx]
                prior = (rescale_nearest(planes0, W, H).contiguous(), rescale_nearest(views0, W, H).contiguous(),
                         rescale_nearest(weak0, W, H).contiguous() if p["use_APD"] else None)
            planes, weak, views = call_backend(W, H, p, [cams[j] for j in order], [imgs[j] for j in order], depths, prior)
            d = planes[..., 3]
            bad = (d < p["depth_min"]) | (d > p["depth_max"])   # main.cpp:109-112 (float32 comparisons)
            d.masked_fill_(bad, 0)      # in place on the strided view; no host synchronisation (unlike d[bad] = 0)
            weak.masked_fill_(bad, 2)
            state[idx] = (planes, weak, views)
            depth_store[idx] = d.contiguous()
            if log:
                log("pass %d (round %d, scale %d) view %d done on rank %d" % (spec.iteration_index, spec.round_index, spec.scale_size, idx, rank))
        if distributed:  # also with one rank under an initialised process group: the collective path is the only path then
            gathered = sharding.allgather_maps({v: depth_store[v][..., None] for v in mine}, V, group=group)
            for v in range(V):
                if v not in state:
                    depth_store[v] = gathered[v, ..., 0].contiguous()

    def to_view_state(planes, weak, views):
        # split depth / normal on the device: the host then receives two contiguous arrays instead of re-packing 16 bytes per pixel
        return ViewState(planes[..., 3].contiguous().cpu().numpy(), planes[..., :3].contiguous().cpu().numpy(),
                         weak.cpu().numpy().astype(np.uint8, copy=False), views.contiguous().cpu().numpy().view(np.uint32))

    # before fusion: everybody gets every view's depth + normal + weak (+ selected views)
    if distributed:
        packed = {}
        for v in mine:
            planes, weak, views = state[v]
            packed[v] = torch.cat([planes[..., 3:4], planes[..., :3], weak[..., None].to(torch.float32),
                                   views.contiguous()[..., None].view(torch.float32)], -1)
        g = sharding.allgather_maps(packed, V, group=group)
        out = {}
        for v in range(V):
            gv = g[v]
            out[v] = to_view_state(torch.cat([gv[..., 1:4], gv[..., 0:1]], -1), gv[..., 4].to(torch.uint8),
                                   gv[..., 5].contiguous().view(torch.int32))
        return out
    return {v: to_view_state(*state[v]) for v in state}


def synthetic_ring(synth, width, height, num_views, num_src, camera_factory, seed=0, textureless=0.0):
    """`num_views` reference views on the generator's camera ring, each paired with its `num_src` nearest neighbours.
    camera_factory(K, R, t, W, H, depth_min, depth_max) builds the backend's camera struct."""
    sc = synth.make_scene(width, height, num_views - 1, seed=seed, textureless=textureless)
    imgs = sc.images_numpy()
    cams = [camera_factory(sc.K[i], sc.R[i], sc.t[i], width, height, sc.depth_min, sc.depth_max) for i in range(num_views)]
    pairs = []
    for i in range(num_views):
        others = sorted((j for j in range(num_views) if j != i), key=lambda j: (abs(j - i), j))
        pairs.append(others[:num_src])
    return MvsScene(cams, [np.ascontiguousarray(im, np.float32) for im in imgs], pairs)


# ------------------------------------------------------------------------------------------------------------------
# dense-folder I/O (the reference's on-disk contract) through the C++ host library
# ------------------------------------------------------------------------------------------------------------------

_host = None


def host_lib():
    """libapd_host.so: ReadCamera / ReadGrayImage / ReadBinMat / WriteBinMat of the drop-in host (host/APD.cpp)."""
    global _host
    if _host is None:
        import ctypes as C
        import os
        from . import lib as product_lib, LIB_PATH
        product_lib()  # libapd_host.so links against libapd_mi355x.so
        L = C.CDLL(os.path.join(os.path.dirname(LIB_PATH), "libapd_host.so"))
        ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.apdhost_read_camera.argtypes = [C.c_char_p, C.c_void_p]
        L.apdhost_read_gray_image.argtypes = [C.c_char_p, ip, ip, fp, C.c_size_t]
        L.apdhost_write_bin_mat.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.apdhost_fuse.restype = C.c_longlong
        L.apdhost_round_num.argtypes = [C.c_int, C.c_int]
        L.apdhost_schedule.argtypes = [C.c_int, C.c_int, ip, C.c_int]
        _host = L
    return _host


def load_dense_folder(folder, camera_type):
    """pair.txt (main.cpp:6-49: sources with score <= 0 dropped), cams/%08d_cam.txt, images/%08d.{jpg,pgm}.
    View i of the returned scene is the i-th entry of pair.txt; `ids` maps it back to the image id."""
    import ctypes as C
    import os
    L = host_lib()
    tok = open(os.path.join(folder, "pair.txt")).read().split()
    n = int(tok[0])
    pos = 1
    ids, src_ids = [], []
    for _ in range(n):
        ids.append(int(tok[pos]))
        m = int(tok[pos + 1])
        pos += 2
        srcs = []
        for _ in range(m):
            sid, score = int(tok[pos]), float(tok[pos + 1])
            pos += 2
            if score > 0.0:
                srcs.append(sid)
        src_ids.append(srcs)
    index_of = {v: i for i, v in enumerate(ids)}
    # Sources without an entry of their own (a subset of the views is reconstructed): loaded as source-only views after the
    # reference views, like the reference loads any id it is given (APD.cpp:419-452).
    extra = []
    for v, srcs in zip(ids, src_ids):
        for s_id in srcs:
            if s_id == v:
                raise ValueError("pair.txt: view %d lists itself as a source" % v)
            if s_id not in index_of:
                index_of[s_id] = len(ids) + len(extra)
                extra.append(s_id)
    def load_view(v):
        cam = camera_type()
        if L.apdhost_read_camera(os.path.join(folder, "cams", "%08d_cam.txt" % v).encode(), C.byref(cam)) != 0:
            raise IOError("cannot read camera %d" % v)
        rows, cols = C.c_int(), C.c_int()
        stem = os.path.join(folder, "images", "%08d" % v).encode()
        if L.apdhost_read_gray_image(stem, C.byref(rows), C.byref(cols), None, 0) != 0:
            raise IOError("cannot read image %d" % v)
        img = np.empty((rows.value, cols.value), np.float32)
        L.apdhost_read_gray_image(stem, C.byref(rows), C.byref(cols), img.ctypes.data_as(C.POINTER(C.c_float)), img.size)
        cam.width, cam.height = cols.value, rows.value
        return cam, img

    # the decoder runs outside the GIL (ctypes) and the host library's image cache is locked: one thread per image
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, max(1, len(ids)))) as pool:
        loaded = list(pool.map(load_view, ids + extra))
    cams = [c for c, _ in loaded]
    imgs = [im for _, im in loaded]
    scene = MvsScene(cams, imgs, [[index_of[s] for s in srcs] for srcs in src_ids])
    scene.ids = ids + extra   # image id of every loaded view; the first scene.num_views are the reference views
    return scene


def save_results(folder, scene, results):
    """<dense>/APD/<%08d>/{depths.dmb, normals.dmb, weak.bin, selected_views.bin} as ProcessProblem writes them
    (main.cpp:117-124), for the fusion step."""
    import os
    L = host_lib()
    ids = getattr(scene, "ids", list(range(scene.num_views)))
    for v, st in results.items():
        d = os.path.join(folder, "APD", "%08d" % ids[v])
        os.makedirs(d, exist_ok=True)
        rows, cols = st.depth.shape
        for name, code, arr in (("depths.dmb", 5, st.depth), ("normals.dmb", 21, st.normal), ("weak.bin", 0, st.weak),
                                ("selected_views.bin", 4, st.views)):
            a = np.ascontiguousarray(arr)
            if L.apdhost_write_bin_mat(os.path.join(d, name).encode(), rows, cols, code, a.ctypes.data) != 0:
                raise IOError("cannot write " + os.path.join(d, name))


def load_colour_images(folder, ids):
    """images/%08d.{jpg,ppm,pgm} as cv::imread(IMREAD_COLOR) returns them: float32 [H, W, 3], blue first (APD.cpp:859)."""
    import ctypes as C
    import os
    L = host_lib()
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.apdhost_read_color_image.argtypes = [C.c_char_p, ip, ip, fp, C.c_size_t]
    def load(v):
        stem = os.path.join(folder, "images", "%08d" % v).encode()
        r, c = C.c_int(), C.c_int()
        # size from the grey read (served from the process cache after load_dense_folder) instead of a second colour decode
        if not os.path.exists(stem.decode() + ".ppm") and L.apdhost_read_gray_image(stem, C.byref(r), C.byref(c), None, 0) == 0:
            pass
        elif L.apdhost_read_color_image(stem, C.byref(r), C.byref(c), None, 0) != 0:
            raise IOError("cannot read image %d of %s" % (v, folder))
        a = np.zeros((r.value, c.value, 3), np.float32)
        if L.apdhost_read_color_image(stem, C.byref(r), C.byref(c), a.ctypes.data_as(fp), a.size) != 0 or a.shape[:2] != (r.value, c.value):
            raise IOError("cannot read image %d of %s" % (v, folder))
        return a

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, max(1, len(ids)))) as pool:
        return list(pool.map(load, ids))


def fuse(scene, results, ply_path, device=0, colour_images=None, block_masks=None):
    """RunFusion (APD.cpp:826-977) on the gathered maps: consistency check and merge into a binary PLY on GPU `device`
    (apd_fuse_views, csrc/apd_fusion.hip).  Every view must be at one resolution per view; images are
    resampled to the depth-map size if it differs (RescaleImageAndCamera, APD.cpp:729-750).  colour_images: optional
    float32 [H, W, 3] arrays (blue, green, red, as load_colour_images returns them) for the point colours; the grey
    images of the scene otherwise (blue = green = red).  block_masks: optional uint8 [H, W] arrays (or None per view), the
    `blocks/mask_<id>.jpg` of APD.cpp:849-853: reference pixels below 128 are not fused.  Returns the number of points."""
    import ctypes as C
    L = host_lib()
    L.apdhost_set_fusion_device(int(device))
    V = scene.num_views
    cam_t = type(scene.cameras[0])
    cams = (cam_t * V)()
    imgs, deps, nors, weaks = [], [], [], []
    rows, cols = (C.c_int * V)(), (C.c_int * V)()
    for v in range(V):
        st = results[v]
        h, w = st.depth.shape
        cam = cam_t.from_buffer_copy(scene.cameras[v])
        img = scene.images[v] if colour_images is None else colour_images[v]
        if img.shape[:2] != (h, w):
            sx = np.float32(w) / np.float32(img.shape[1])
            sy = np.float32(h) / np.float32(img.shape[0])
            if img.ndim == 3:
                img = np.stack([np.rint(resize_linear(img[..., k], w, h)) for k in range(3)], -1).astype(np.float32)
            else:
                img = np.rint(resize_linear(img, w, h)).astype(np.float32)
            cam.K[0] = float(np.float32(cam.K[0]) * sx)
            cam.K[2] = float(np.float32(cam.K[2]) * sx)
            cam.K[4] = float(np.float32(cam.K[4]) * sy)
            cam.K[5] = float(np.float32(cam.K[5]) * sy)
        cams[v] = cam
        rows[v], cols[v] = h, w
        imgs.append(np.ascontiguousarray(img, np.float32))
        deps.append(np.ascontiguousarray(st.depth, np.float32))
        nors.append(np.ascontiguousarray(st.normal, np.float32))
        weaks.append(np.ascontiguousarray(rescale_nearest(st.weak, w, h), np.uint8))
    offs = (C.c_int * (V + 1))()
    flat = []
    for v in range(V):
        offs[v] = len(flat)
        flat += [s for s in scene.pairs[v] if s < V]  # source-only views have no maps to check against
    offs[V] = len(flat)
    idx = (C.c_int * max(len(flat), 1))(*flat)

    def ptrs(arrs):
        return (C.c_void_p * V)(*[a.ctypes.data for a in arrs])

    channels = 3 if imgs[0].ndim == 3 else 1
    blocks = None
    if block_masks is not None:
        keep = [None if b is None else np.ascontiguousarray(b, np.uint8) for b in block_masks]
        blocks = (C.c_void_p * V)(*[None if b is None else b.ctypes.data for b in keep])
    n = L.apdhost_fuse(V, C.byref(cams), ptrs(imgs), channels, ptrs(deps), ptrs(nors), ptrs(weaks), blocks, rows, cols, offs, idx,
                       str(ply_path).encode())
    if n < 0:
        raise RuntimeError("device fusion failed (apd_fuse_views): see stderr")
    return int(n)
