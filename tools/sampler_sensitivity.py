#!/usr/bin/env python3
"""How far can a physical run of the reference end from the contract's bits?  (CPU only, oracle only.)

The reference samples its source images with the CUDA texture unit (APD.cpp:598-602), whose bilinear weights are 9-bit
fixed point with 8 fractional bits; contract C7 uses float weights, because a fixed-point weight is not something
the reference's authors chose, it is what their hardware does.  This tool runs the reference's three pass kinds twice on
the oracle -- contract arithmetic, and the same with both weights rounded to the nearest 1/256 (oracle study knob
orc_set_study_weights_q8) -- and prints, for the final planes, the statistics tests/test_gpu_fast_rcp.py prints for the
fast-reciprocal mode: the fraction of pixels inside north_star's per-pixel tolerance (1e-3 relative depth AND 1 degree), and
both runs' accuracy against the ground truth.

    python tools/sampler_sensitivity.py [WxH] [num_src]        (default 640x480, 5 sources; about ten minutes on 8 cores)
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from oracle import binding as ob  # noqa: E402

synth = importlib.import_module("apd-mvs_amd.synth")


def three_passes(sc, imgs, N, deps):
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.0075, geom_consistency=1)]
    prior = None
    for extra in passes:
        p = common.base_params(sc, N, seed=99, max_iterations=3, **extra)
        o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if p.get("geom_consistency") else None, prior=prior)
        o.run()
        prior = common.postprocess(o.planes.copy(), o.weak_info.copy(), o.selected_views.copy(), p["depth_min"], p["depth_max"])
        o.close()
    return prior[0], prior[2]


def main():
    wh = sys.argv[1] if len(sys.argv) > 1 else "640x480"
    W, H = (int(v) for v in wh.split("x"))
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=8, textureless=0.15)
    gt = sc.gt_depth.numpy()
    deps = [gt.copy() for _ in range(N + 1)]
    L = ob.lib()
    t0 = time.time()
    L.orc_set_study_weights_q8(0)
    exact, weak_e = three_passes(sc, imgs, N, deps)
    again, _ = three_passes(sc, imgs, N, deps)
    assert np.array_equal(exact.view(np.uint32), again.view(np.uint32)), "the oracle must be reproducible run to run"
    L.orc_set_study_weights_q8(1)
    try:
        q8, weak_q = three_passes(sc, imgs, N, deps)
    finally:
        L.orc_set_study_weights_q8(0)
    de, dq = exact[..., 3].astype(np.float64), q8[..., 3].astype(np.float64)
    both = (de > 0) & (dq > 0)
    rel = (np.abs(de - dq) / np.where(de > 0, de, 1))[both]
    cosang = np.clip((exact[..., :3].astype(np.float64) * q8[..., :3]).sum(-1), -1, 1)
    ang = np.degrees(np.arccos(cosang))[both]
    inside = ((rel <= 1e-3) & (ang <= 1.0)).mean()
    q = lambda d: float((np.abs(d - gt) / gt < 0.01)[8:-8, 8:-8].mean())
    e = lambda d: float(np.median((np.abs(d - gt) / gt)[8:-8, 8:-8]))
    print("sampler sensitivity, %dx%d, %d sources, three pass kinds (FIRST_INIT, REFINE_INIT + APD, REFINE_ITER + geometric), %.0f s" % (
        W, H, N, time.time() - t0))
    print("  contract (float weights) vs 8-bit fractional weights (CUDA texture unit):")
    print("  identical bits: %.4f of the plane components" % float((exact.view(np.uint32) == q8.view(np.uint32)).mean()))
    print("  pixels valid in both: %.4f; of those within 1e-3 depth AND 1 degree (north_star's per-pixel tolerance): %.4f" % (
        float(both.mean()), float(inside)))
    print("  depth alone within 1e-3: %.4f, 3e-3: %.4f, 1e-2: %.4f (median %.2e)" % (
        float((rel <= 1e-3).mean()), float((rel <= 3e-3).mean()), float((rel <= 1e-2).mean()), float(np.median(rel))))
    print("  normals within 1 deg: %.4f, 5 deg: %.4f (median %.2f deg)" % (float((ang <= 1).mean()), float((ang <= 5).mean()), float(np.median(ang))))
    print("  validity agrees on %.4f of the pixels; WEAK maps equal on %.4f" % (float(((de > 0) == (dq > 0)).mean()), float((weak_e == weak_q).mean())))
    print("  against the ground truth: within 1 %% of depth: contract %.4f, 8-bit weights %.4f; median relative error %.2e / %.2e" % (
        q(de), q(dq), e(de), e(dq)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
