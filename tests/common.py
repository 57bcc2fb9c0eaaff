"""Shared scene / handle / oracle construction for the tests."""
import numpy as np


def scene_inputs(synth, W, H, N, seed=1, textureless=0.0, rotate=True):
    sc = synth.make_scene(W, H, N, seed=seed, textureless=textureless, rotate=rotate)
    return sc, sc.images_numpy()


def base_params(sc, N, **kw):
    p = dict(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0, state=0,
             max_iterations=2, seed=7)
    p.update(kw)
    return p


def make_oracle(ob, sc, imgs, N, params, depths=None, prior=None):
    W, H = sc.width, sc.height
    cams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    pr = prior or (None, None, None)
    return ob.Oracle(W, H, ob.default_params(**params), cams, imgs, depths=depths, prior_planes=pr[0], prior_views=pr[1],
                     prior_weak=pr[2])


def make_handle(pkg, sc, imgs, N, params, depths=None, prior=None, device=0, options=None):
    W, H = sc.width, sc.height
    cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    h = pkg.Handle(W, H, pkg.default_params(**params), device=device)
    for name, value in (options or {}).items():  # apd_set_option, before the upload latches the upload-time ones
        h.set_option(name, value)
    h.upload_views(cams, imgs, depths)
    if prior is not None:
        h.upload_prior(*prior)
    return h


ORACLE_STATES = [("planes", "STATE_PLANES", "planes"), ("costs", "STATE_COSTS", "costs"), ("rng", "STATE_RNG", "rng"),
                 ("views", "STATE_SELECTED_VIEWS", "selected_views"), ("view_weight", "STATE_VIEW_WEIGHT", "view_weight"),
                 ("weak", "STATE_WEAK_INFO", "weak_info"), ("fit", "STATE_FIT_PLANES", "fit_planes"),
                 ("reliable", "STATE_WEAK_RELIABLE", "weak_reliable"), ("nearest", "STATE_NEAREST_STRONG", "nearest_strong"),
                 ("neighbours", "STATE_NEIGHBOURS", "neighbours")]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def assert_state_equal(pkg, h, o, where, skip=()):
    """Bit-exact comparison of every state array (integer / byte / index work and float32 alike)."""
    for name, hs, oa in ORACLE_STATES:
        if name in skip or (name == "neighbours" and h.weak_count == 0):
            continue
        a, b = h.state(getattr(pkg, hs)), getattr(o, oa)
        if not np.array_equal(bits(a), bits(b)):
            neq = bits(a).reshape(a.shape[0], -1) != bits(b).reshape(a.shape[0], -1)
            raise AssertionError("%s: HIP and oracle differ in `%s` (%d bytes differ)" % (where, name, int(neq.sum())))


def postprocess(planes, weak, views, dmin, dmax, unknown=2):
    """ProcessProblem post-processing (main.cpp:105-115): out-of-range depth -> 0 and UNKNOWN."""
    planes, weak = planes.copy(), weak.copy()
    d = planes[..., 3]
    bad = (d < dmin) | (d > dmax)
    d[bad] = 0
    weak[bad] = unknown
    return planes, views.copy(), weak


def depth_of_planes(planes, K):
    """ComputeDepthfromPlaneHypothesis (APD.cu:206-209) in float64, for quality metrics only."""
    H, W = planes.shape[:2]
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    pl = planes.astype(np.float64)
    return -pl[..., 3] * K[0] / ((xs - K[2]) * pl[..., 0] + (K[0] / K[4]) * (ys - K[5]) * pl[..., 1] + K[0] * pl[..., 2])


def fake_depth_maps(W, H, n):
    ys, xs = np.mgrid[0:H, 0:W]
    out = []
    for k in range(n):
        d = (2.2 + 0.1 * np.sin(0.05 * xs + k) + 0.05 * np.cos(0.07 * ys)).astype(np.float32)
        d[(xs % 17 == 0) & (ys % 13 == 0)] = 0.0
        out.append(d)
    return out


# ---- full-resolution lock-step against the oracle's region-of-interest mode ----------------------------------------------

def download_all(pkg, h):
    """Every state array of the handle as numpy, keyed like ORACLE_STATES."""
    out = {}
    for name, hs, _ in ORACLE_STATES:
        if name == "neighbours" and h.weak_count == 0:
            continue
        out[name] = h.state(getattr(pkg, hs))
    return out


def load_into_oracle(o, snap):
    """Overwrites the oracle's state arrays with a snapshot of the HIP state (the pre-kernel state of the next step)."""
    for name, _, oa in ORACLE_STATES:
        if name in snap:
            getattr(o, oa)[...] = snap[name]


def assert_windows_equal(o, snap, windows, where, weak_before=None):
    """HIP snapshot vs oracle, bit for bit, on every window (x0, y0, x1, y1).  The neighbour table is indexed by WEAK pixel:
    its rows are compared for the pixels of the window that were WEAK when the table was allocated (`weak_before`)."""
    nmap = o.neighbours_map
    for (x0, y0, x1, y1) in windows:
        for name, _, oa in ORACLE_STATES:
            if name not in snap:
                continue
            a, b = snap[name], getattr(o, oa)
            if name == "neighbours":
                if weak_before is None:
                    continue
                rows = nmap[y0:y1, x0:x1][weak_before[y0:y1, x0:x1] == 0]
                a, b = a[rows], b[rows]
            else:
                a, b = a[y0:y1, x0:x1], b[y0:y1, x0:x1]
            if not np.array_equal(bits(a), bits(b)):
                diff = (bits(a).reshape(a.shape[0], -1) != bits(b).reshape(a.shape[0], -1)).sum()
                raise AssertionError("%s: HIP and oracle differ in `%s` on window %s (%d bytes)" % (where, name, (x0, y0, x1, y1), int(diff)))


def fullsize_lockstep(pkg, h, o, schedule, windows, label, log=None):
    """The HIP path runs every kernel of `schedule` over the whole image; the oracle runs the same kernel on the windows
    only, from the HIP path's own pre-kernel state (copied in once per kernel), and the windows are compared bit for bit
    after every kernel.  Launches never read what they write (red/black colouring), so a window's result does not depend on
    what happens outside it in the same launch (tests/test_oracle_roi.py checks that property of the oracle)."""
    import time
    # kernels read-modify-write their own pixels (RNG state): the oracle must visit every pixel once, so overlapping windows
    # are dropped (later ones lose)
    disjoint = []
    for w in windows:
        w = tuple(int(v) for v in w)
        if all(w[2] <= d[0] or d[2] <= w[0] or w[3] <= d[1] or d[3] <= w[1] for d in disjoint):
            disjoint.append(w)
    assert len(disjoint) >= max(1, len(windows) - 2), (windows, disjoint)
    windows = disjoint
    snap = download_all(pkg, h)
    load_into_oracle(o, snap)
    weak_alloc = snap["weak"].copy()  # the WEAK map the neighbour table was laid out for (before K4 demotes pixels)
    compared = 0
    for kid, it in schedule:
        t0 = time.perf_counter()
        h.run_kernel(kid, it)
        t1 = time.perf_counter()
        for (x0, y0, x1, y1) in windows:
            o.set_roi(x0, y0, x1, y1)
            o.run_kernel(kid, it)
        o.set_roi()
        t2 = time.perf_counter()
        snap = download_all(pkg, h)
        assert_windows_equal(o, snap, windows, "%s after K%d(iter %d)" % (label, kid, it), weak_alloc)
        load_into_oracle(o, snap)
        compared += 1
        if log is not None:
            log.append("K%d(iter %d): HIP %.0f ms, oracle windows %.1f s, download+compare+sync %.1f s"
                       % (kid, it, (t1 - t0) * 1e3, t2 - t1, time.perf_counter() - t2))
    return compared


class OracleBackend:
    """The CPU oracle behind the pipeline's backend interface (tests only): one (view, pass) == orc_run."""
    device = None

    def __init__(self, threads=2):
        from oracle import binding as ob
        self.ob = ob
        ob.lib().orc_set_threads(threads)

    @property
    def camera_type(self):
        return self.ob.Camera

    def run_pass(self, width, height, params, cameras, images, depths, prior):
        ob = self.ob
        pr = prior or (None, None, None)
        o = ob.Oracle(width, height, ob.default_params(**params), cameras, images, depths=depths, prior_planes=pr[0],
                      prior_views=pr[1], prior_weak=pr[2])
        o.run()
        out = o.planes.copy(), o.weak_info.copy(), o.selected_views.copy()
        o.close()
        return out
