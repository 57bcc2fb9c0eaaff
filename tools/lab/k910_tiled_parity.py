"""Lab check for the -DAPD_K910_SUBPATCH_TILED=1 build: three passes (FIRST_INIT, REFINE_INIT + APD, REFINE_ITER + APD + geometric
term) with --opt tiled_copy=2 against the oracle, every state array bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
from oracle import binding as ob
import common
for (W, H, N, seed) in ((160, 120, 4, 3), (233, 141, 10, 5)):
    sc, imgs = common.scene_inputs(synth, W, H, N, seed=seed, textureless=0.25)
    deps = common.fake_depth_maps(W, H, N + 1)
    passes = [dict(state=0, use_APD=0, weak_peak_radius=6),
              dict(state=1, use_APD=1, weak_peak_radius=6, rotate_time=2, ransac_threshold=0.00875),
              dict(state=2, use_APD=1, weak_peak_radius=4, rotate_time=4, ransac_threshold=0.0075, geom_consistency=1)]
    prior = None
    for pi, extra in enumerate(passes):
        p = common.base_params(sc, N, seed=11, **extra)
        geom = bool(p.get("geom_consistency"))
        h = common.make_handle(pkg, sc, imgs, N, p, depths=deps if geom else None, prior=prior, options={"tiled_copy": 2})
        o = common.make_oracle(ob, sc, imgs, N, p, depths=deps if geom else None, prior=prior)
        h.run(); o.run()
        common.assert_state_equal(pkg, h, o, "%dx%d N=%d pass %d" % (W, H, N, pi))
        planes, weak, views = h.download()
        prior = common.postprocess(planes, weak, views, p["depth_min"], p["depth_max"])
        print("ok %dx%d N=%d pass %d weak %d" % (W, H, N, pi, h.weak_count))
        h.close(); o.close()
