"""Loads a tests/golden/*.npz fixture into the pieces the oracle / HIP handle constructors need."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["first_pass_48x36", "apd_pass_64x48", "geom_pass_64x48"]
INT_KEYS = {"max_iterations", "num_images", "top_k", "geom_consistency", "use_APD", "weak_peak_radius", "rotate_time",
            "state", "seed"}


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        self.z = z
        self.W, self.H, self.N = int(z["width"]), int(z["height"]), int(z["num_src"])
        self.imgs = [im.astype(np.float32) for im in z["images_u8"]]
        self.K, self.R, self.t = list(z["K"]), list(z["R"]), list(z["t"])
        self.depth_min, self.depth_max = [float(v) for v in z["cam_depth_range"]]
        self.params = {}
        for k, v in zip(z["param_keys"], z["param_values"]):
            self.params[str(k)] = int(v) if str(k) in INT_KEYS else float(v)
        self.depths = [d for d in z["depths_bits"].view(np.float32)] if "depths_bits" in z else None
        self.prior = None
        if "prior_planes_bits" in z:
            self.prior = (z["prior_planes_bits"].view(np.float32), z["prior_views"], z["prior_weak"])

    def cameras(self, mod):
        return [mod.make_camera(self.K[i], self.R[i], self.t[i], self.W, self.H, self.depth_min, self.depth_max)
                for i in range(self.N + 1)]

    def check(self, planes, costs, views, weak, view_weight, rng, neighbours=None):
        z = self.z
        assert np.array_equal(np.ascontiguousarray(planes).view(np.uint32), z["out_planes_bits"]), "planes"
        assert np.array_equal(np.ascontiguousarray(costs).view(np.uint32), z["out_costs_bits"]), "costs"
        assert np.array_equal(views, z["out_selected_views"]), "selected views"
        assert np.array_equal(weak, z["out_weak_info"]), "weak info"
        assert np.array_equal(view_weight, z["out_view_weight"]), "view weights"
        assert np.array_equal(rng, z["out_rng"]), "rng"
        if neighbours is not None and "out_neighbours" in z:
            assert np.array_equal(neighbours, z["out_neighbours"]), "neighbours"
