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

if __name__ == "__main__":
    ok = run(64, 48, 3)
    ok &= run(97, 71, 5, iters=2)
    print("ALL OK" if ok else "SOME DIFF")
