"""Randomised bit-exact parity sweep: HIP path vs CPU oracle over random sizes, view counts, scenes and seeds through the
three pass kinds (FIRST_INIT, REFINE_INIT + APD, REFINE_ITER + APD + geometric term), 8-bit and float images, compared
after every pass (all state arrays).  Usage: [APD_FUZZ_SCALE=3] [APD_FUZZ_HARD=1] python tools/parity_fuzz.py [cases] [first_seed]
APD_FUZZ_HARD=1: every case on synth.HARD-like scenes (slabs in front of the planes: depth steps and occlusions; per-view gain /
offset; sources aiming off the target), with the clutter, gain and aim drawn per case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
from oracle import binding as ob
import common


def run_case(case, hard=None):
    """One random configuration through the three pass kinds; raises AssertionError on the first differing state array."""
    rng = np.random.RandomState(1000 + case)
    if hard is None:
        hard = os.environ.get("APD_FUZZ_HARD", "0") == "1"
    scale = float(os.environ.get("APD_FUZZ_SCALE", "1"))  # larger frames: more tiles, supertiles and list blocks per launch (the oracle takes scale^2 longer)
    W, H = int(rng.randint(36, 260) * scale), int(rng.randint(30, 180) * scale)
    N = int(rng.randint(1, 10))
    if rng.rand() < 0.25:  # ten and more sources: other kernel instantiations (NMAX = 12 / 16 / 32, K14 walking (sample, lane) pairs)
        N = int(rng.randint(10, 19))
    tl = float(rng.choice([0.0, 0.15, 0.3]))
    iters = int(rng.randint(1, 4))
    float_images = bool(rng.rand() < 0.35)
    rotate = bool(rng.rand() < 0.8)
    scene_kw = {}
    if hard:
        hr = np.random.RandomState(77000 + case)   # its own stream: the easy cases keep their draws
        scene_kw = dict(clutter=int(hr.randint(3, 20)), gain=float(hr.choice([0.0, 0.05, 0.1, 0.2])), baseline=float(hr.choice([0.06, 0.1, 0.14])),
                        aim_jitter=float(hr.choice([0.0, 0.2, 0.4, 0.7])))
    sc = synth.make_scene(W, H, N, seed=case, textureless=tl, rotate=rotate, **scene_kw)
    imgs = sc.images_numpy()
    if float_images:  # what a resampled pyramid level holds: non-integer grey values
        imgs = [(im * np.float32(0.731) + np.float32(3.3) * np.sin(np.arange(im.size, dtype=np.float32).reshape(im.shape) * 0.01)).astype(np.float32)
                for im in imgs]
    deps = common.fake_depth_maps(W, H, N + 1)
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.01 - 0.00125),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.01 - 0.0025, geom_consistency=1)]
    if rng.rand() < 0.5:  # the schedule's other radii (main.cpp:176-186: 6, 4, 2, 2) and values no schedule uses (K14's peak test, its end samples)
        for q in passes:
            q["weak_peak_radius"] = int(rng.choice([2, 2, 2, 4, 0, 1, 3, 5, 12, 29, 30, 40]))
    prior = None
    # how the HIP side is driven (the oracle always runs the plain schedule): one handle per pass (the reference's object per
    # view and pass), or ONE handle recycled with apd_reset for all three passes; the whole pass in one call, or split around the
    # depth maps (apd_upload_views_split / apd_run_before_depths / apd_upload_depths / apd_run_after_depths) as the scheduler
    # with several views in flight drives it
    recycle = bool(rng.rand() < 0.5)
    split = bool(rng.rand() < 0.5)
    share = bool(split and rng.rand() < 0.5)   # images created once on the device (apd_image_create) and uploaded by reference
    label = "case %d: %dx%d N=%d textureless=%.2f iters=%d radii=%s %s%s%s%s" % (case, W, H, N, tl, iters, "/".join(str(q["weak_peak_radius"]) for q in passes),
                                                                               "float" if float_images else "8-bit", " recycled" if recycle else "",
                                                                               (" shared" if share else " split") if split else "",
                                                                               (" hard(clutter %d gain %.2f baseline %.2f aim %.1f)" % (
                                                                                   scene_kw["clutter"], scene_kw["gain"], scene_kw["baseline"], scene_kw["aim_jitter"])) if hard else "")
    cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    shared = [pkg.SharedImage(W, H, im) for im in imgs] if share else None
    h = None
    try:
        for pi, extra in enumerate(passes):
            p = common.base_params(sc, N, seed=100 + case, max_iterations=iters, **extra)
            geom = bool(p.get("geom_consistency"))
            if h is None:
                h = pkg.Handle(W, H, pkg.default_params(**p), device=0)
            else:
                h.reset(pkg.default_params(**p))
            if share:
                h.upload_views_shared(cams, shared)
            elif split:
                h.upload_views_split(cams, imgs)
            else:
                h.upload_views(cams, imgs, deps if geom else None)
            if prior is not None:
                h.upload_prior(*prior)
            o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
            try:
                assert h.weak_count == o.weak_count, label
                if split:
                    h.run_before_depths()
                    if geom:
                        h.upload_depths(deps)
                    h.run_after_depths()
                else:
                    h.run()
                o.run()
                common.assert_state_equal(pkg, h, o, "%s pass %d" % (label, pi))
                planes, weak, views = h.download()
                prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
            finally:
                o.close()
            if not recycle:
                h.close()
                h = None
    finally:
        if h is not None:
            h.close()
        for im in shared or []:
            im.close()
    return label


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t0 = time.time()
    for case in range(first, first + cases):
        try:
            print("ok   " + run_case(case), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL case %d: %s" % (case, e), flush=True)
    print("%d case(s), %d failure(s), %.0f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
