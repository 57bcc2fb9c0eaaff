#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs (8-bit images, cameras, parameters, prior state) and the
oracle's outputs after a full pass of the RunPatchMatch schedule.

The reference has no fixtures and cannot be built here, so these vectors pin the ORACLE (regression
anchor for both the oracle and the HIP path), not the reference.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
import common  # noqa: E402

PARAM_KEYS = ["max_iterations", "num_images", "top_k", "depth_min", "depth_max", "geom_consistency", "use_APD",
              "weak_peak_radius", "rotate_time", "ransac_threshold", "geom_factor", "state", "seed"]


def save_case(name, sc, imgs, N, params, o, depths=None, prior=None):
    out = {
        "width": sc.width, "height": sc.height, "num_src": N,
        "images_u8": np.stack(imgs).astype(np.uint8),
        "K": np.stack(sc.K), "R": np.stack(sc.R), "t": np.stack(sc.t),
        "cam_depth_range": np.array([sc.depth_min, sc.depth_max], np.float32),
        "param_keys": np.array(PARAM_KEYS),
        "param_values": np.array([float(params.get(k, getattr(o.params, k))) for k in PARAM_KEYS], np.float64),
        "out_planes_bits": o.planes.view(np.uint32).copy(),
        "out_costs_bits": o.costs.view(np.uint32).copy(),
        "out_selected_views": o.selected_views.copy(),
        "out_weak_info": o.weak_info.copy(),
        "out_view_weight": o.view_weight.copy(),
        "out_rng": o.rng.copy(),
    }
    if depths is not None:
        out["depths_bits"] = np.stack(depths).view(np.uint32)
    if prior is not None:
        out["prior_planes_bits"] = prior[0].view(np.uint32)
        out["prior_views"] = prior[1]
        out["prior_weak"] = prior[2]
        out["out_neighbours"] = o.neighbours.copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    pkg = ge.load_package()
    from apd_mvs_amd import synth
    from oracle import binding as ob

    # A: first pass (FIRST_INIT), 48x36, 3 source views
    W, H, N = 48, 36, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=1)
    pA = common.base_params(sc, N, seed=2024, max_iterations=2, weak_peak_radius=6)
    o = common.make_oracle(ob, sc, imgs, N, pA)
    o.run()
    save_case("first_pass_48x36", sc, imgs, N, pA, o)

    # B: APD pass (REFINE_INIT) on a scene with textureless rectangles, 64x48
    W, H, N = 64, 48, 3
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.3)
    p1 = common.base_params(sc, N, seed=31, max_iterations=2, weak_peak_radius=6)
    o1 = common.make_oracle(ob, sc, imgs, N, p1)
    o1.run()
    prior = common.postprocess(o1.planes, o1.weak_info, o1.selected_views, p1["depth_min"], p1["depth_max"])
    pB = common.base_params(sc, N, seed=31, max_iterations=2, weak_peak_radius=6, state=ob.REFINE_INIT, use_APD=1,
                            rotate_time=2, ransac_threshold=0.01 - 0.00125)
    o2 = common.make_oracle(ob, sc, imgs, N, pB, prior=prior)
    o2.run()
    print("  case B weak pixels:", o2.weak_count)
    save_case("apd_pass_64x48", sc, imgs, N, pB, o2, prior=prior)

    # C: geometric pass (REFINE_ITER + geom_consistency) continuing from B
    prior2 = common.postprocess(o2.planes, o2.weak_info, o2.selected_views, pB["depth_min"], pB["depth_max"])
    deps = common.fake_depth_maps(W, H, N + 1)
    pC = common.base_params(sc, N, seed=31, max_iterations=2, weak_peak_radius=4, state=ob.REFINE_ITER, use_APD=1,
                            rotate_time=4, ransac_threshold=0.01 - 0.0025, geom_consistency=1)
    o3 = common.make_oracle(ob, sc, imgs, N, pC, depths=deps, prior=prior2)
    o3.run()
    print("  case C weak pixels:", o3.weak_count)
    save_case("geom_pass_64x48", sc, imgs, N, pC, o3, depths=deps, prior=prior2)


if __name__ == "__main__":
    main()
