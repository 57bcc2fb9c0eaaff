"""Ad-hoc kernel-by-kernel parity probe (HIP vs oracle). Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
from oracle import binding as ob

def cmp(name, a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    same = np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if not same:
        neq = (a != b)
        if a.dtype.kind == 'f':
            neq = a.view(np.uint32) != b.view(np.uint32)
        idx = np.argwhere(neq)
        print("   MISMATCH %s: %d / %d elements; first at %s  hip=%s orc=%s" % (name, neq.sum(), neq.size, idx[0], a[tuple(idx[0])], b[tuple(idx[0])]))
    return same

def run(W, H, N, iters=2, seed=7, textureless=0.0, **pk):
    sc = synth.make_scene(W, H, N, seed=1, textureless=textureless)
    imgs = sc.images_numpy()
    kw = dict(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0, state=pkg.FIRST_INIT, max_iterations=iters, seed=seed)
    kw.update(pk)
    cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    ocams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    h = pkg.Handle(W, H, pkg.default_params(**kw), device=0)
    h.upload_views(cams, imgs)
    o = ob.Oracle(W, H, ob.default_params(**kw), ocams, imgs)
    sched = [1, 2, 5] + [k for i in range(iters) for k in (6, 7, 8)] + [11, 12, 13, 14, 15]
    it = 0; ok_all = True
    for kid in sched:
        t0 = time.time(); h.run_kernel(kid, it); t1 = time.time(); o.run_kernel(kid, it); t2 = time.time()
        ok = True
        ok &= cmp("planes", h.state(pkg.STATE_PLANES), o.planes)
        ok &= cmp("costs", h.state(pkg.STATE_COSTS), o.costs)
        ok &= cmp("rng", h.state(pkg.STATE_RNG), o.rng)
        ok &= cmp("views", h.state(pkg.STATE_SELECTED_VIEWS), o.selected_views)
        ok &= cmp("view_weight", h.state(pkg.STATE_VIEW_WEIGHT), o.view_weight)
        ok &= cmp("weak", h.state(pkg.STATE_WEAK_INFO), o.weak_info)
        ok &= cmp("fit", h.state(pkg.STATE_FIT_PLANES), o.fit_planes)
        print("K%-2d it=%d %s  hip %.3fs  oracle %.3fs" % (kid, it, "OK" if ok else "DIFF", t1 - t0, t2 - t1), flush=True)
        ok_all &= ok
        if kid == 8: it += 1
    gt = sc.gt_depth.numpy(); d = h.state(pkg.STATE_PLANES)[..., 3]
    print("within 1%%: %.4f" % ((np.abs(d - gt) / gt)[8:-8, 8:-8] < 0.01).mean(), "states", np.bincount(h.state(pkg.STATE_WEAK_INFO).ravel(), minlength=3))
    return ok_all

STATES = [("planes", "STATE_PLANES", "planes"), ("costs", "STATE_COSTS", "costs"), ("rng", "STATE_RNG", "rng"),
          ("views", "STATE_SELECTED_VIEWS", "selected_views"), ("view_weight", "STATE_VIEW_WEIGHT", "view_weight"),
          ("weak", "STATE_WEAK_INFO", "weak_info"), ("fit", "STATE_FIT_PLANES", "fit_planes"),
          ("reliable", "STATE_WEAK_RELIABLE", "weak_reliable"), ("nearest", "STATE_NEAREST_STRONG", "nearest_strong"),
          ("neighbours", "STATE_NEIGHBOURS", "neighbours")]

def run_multipass(W, H, N, textureless=0.25, seed=11):
    """pass 1 FIRST_INIT -> pass 2 REFINE_INIT + APD -> pass 3 REFINE_ITER + APD + geometric term."""
    sc = synth.make_scene(W, H, N, seed=3, textureless=textureless)
    imgs = sc.images_numpy()
    cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    ocams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    base = dict(num_images=N + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, max_iterations=2, seed=seed)
    ok_all = True
    prior = None
    ys, xs = np.mgrid[0:H, 0:W]
    fake_depths = [(2.2 + 0.1 * np.sin(0.05 * xs + k) + 0.05 * np.cos(0.07 * ys)).astype(np.float32) for k in range(N + 1)]
    for d in fake_depths:
        d[(xs % 17 == 0) & (ys % 13 == 0)] = 0.0
    passes = [dict(state=pkg.FIRST_INIT, use_APD=0, weak_peak_radius=6),
              dict(state=pkg.REFINE_INIT, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.01 - 0.00125),
              dict(state=pkg.REFINE_ITER, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.01 - 0.0025, geom_consistency=1)]
    for pi, extra in enumerate(passes):
        kw = dict(base); kw.update(extra)
        geom = kw.get("geom_consistency", 0)
        h = pkg.Handle(W, H, pkg.default_params(**kw), device=0)
        h.upload_views(cams, imgs, fake_depths if geom else None)
        o = ob.Oracle(W, H, ob.default_params(**kw), ocams, imgs, depths=fake_depths if geom else None,
                      prior_planes=None if prior is None else prior[0], prior_views=None if prior is None else prior[1],
                      prior_weak=None if prior is None else prior[2])
        if prior is not None:
            h.upload_prior(*prior)
        print("== pass %d: state=%d weak_count hip=%d orc=%d geom=%d" % (pi, kw["state"], h.weak_count, o.weak_count, geom))
        iters = kw["max_iterations"]
        sched = [1, 2, 3, 4, 5] + [k for i in range(iters) for k in (6, 7, 8, 9, 10)] + [11, 12, 13, 14, 15]
        it = 0
        for kid in sched:
            if kid in (3, 4, 9, 10) and h.weak_count == 0:
                continue
            t0 = time.time(); h.run_kernel(kid, it); t1 = time.time(); o.run_kernel(kid, it); t2 = time.time()
            ok = True
            for name, hs, oa in STATES:
                if name == "neighbours" and h.weak_count == 0:
                    continue
                ok &= cmp(name, h.state(getattr(pkg, hs)), getattr(o, oa))
            print("K%-2d it=%d %s  hip %.3fs  oracle %.3fs" % (kid, it, "OK" if ok else "DIFF", t1 - t0, t2 - t1), flush=True)
            ok_all &= ok
            if kid == 10 or (kid == 8 and h.weak_count == 0): it += 1
        # ProcessProblem post-processing (main.cpp:105-115)
        planes, weak, views = o.planes.copy(), o.weak_info.copy(), o.selected_views.copy()
        d = planes[..., 3]
        bad = (d < kw["depth_min"]) | (d > kw["depth_max"])
        planes[..., 3][bad] = 0; weak[bad] = pkg.UNKNOWN
        print("   states after pass:", np.bincount(weak.ravel(), minlength=3))
        prior = (planes, views, weak)
        h.close(); o.close()
    return ok_all

if __name__ == "__main__":
    ok = run_multipass(96, 72, 4)
    ok &= run(64, 48, 3)
    ok &= run(97, 71, 5, iters=2)
    print("ALL OK" if ok else "SOME DIFF")
