#!/usr/bin/env python3
"""Golden vectors for tools/colmap2mvsnet.py, made by RUNNING THE REFERENCE'S OWN CONVERTER in this container.

    python tests/golden/make_colmap_golden.py        (needs /root/reference; not needed to run the tests)

What is pinned: everything `processing_single_scene` (`/root/reference/colmap2mvsnet.py:304-456`) computes and writes before
it touches an image -- cams/%08d_cam.txt (extrinsics, intrinsics, depth range line) and pair.txt (view selection) -- for a
text model with default arguments and for the same model in COLMAP's binary format with --max_d 0 (inverse-depth sample
count), --scale_factor 2, --interval_scale 0.8.  The inputs are the synthetic sparse model of tests/test_colmap_converter.py
(non-contiguous image ids, two camera models, one degenerate-baseline pair).

Two things the reference script needs that this image does not have, and how they are handled -- neither takes part in any
number that is stored:
  * `import cv2` (`:17`).  cv2 is only used for the image conversion at the very end (`:454-469`), after cams/ and pair.txt are
    on disk.  An EMPTY module object stands in for the import; the first cv2 call raises AttributeError, which ends the run.
    The image conversion (pad, nearest-neighbour resize, JPEG) therefore stays unpinned.
  * `np.asscalar` (`:376`), removed in numpy 1.23.  It is given back as `lambda a: a.item()`, the replacement numpy's own
    deprecation note prescribes.
The reference's source is executed where it lies; nothing of it is copied here.  Outputs: tests/golden/colmap/<case>/."""
import argparse
import importlib.util
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/colmap2mvsnet.py"
OUT = os.path.join(HERE, "colmap")

CASES = {
    "text_default": dict(model_ext=".txt", max_d=192, interval_scale=1, scale_factor=1),
    "binary_inverse_depth_scale2": dict(model_ext=".bin", max_d=0, interval_scale=0.8, scale_factor=2),
}


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # never called for anything that is stored (see above)
    if not hasattr(np, "asscalar"):
        np.asscalar = lambda a: a.item()
    spec = importlib.util.spec_from_file_location("ref_colmap2mvsnet", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod  # its worker pool pickles calc_score by module name
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_colmap_converter as tcc  # the synthetic model and its writers
    ref = load_reference()
    cams, views, point_ids, pts = tcc._scene()
    for name, a in CASES.items():
        case = os.path.join(OUT, name)
        shutil.rmtree(case, ignore_errors=True)
        dense = os.path.join(case, "input")
        model = os.path.join(dense, "dslr_calibration_undistorted")
        (tcc._write_text if a["model_ext"] == ".txt" else tcc._write_binary)(model, cams, views, point_ids, pts)
        os.makedirs(os.path.join(dense, "images"))
        save = os.path.join(case, "expected")
        os.makedirs(save)
        args = argparse.Namespace(dense_folder=dense, save_folder=save, theta0=5, sigma1=1, sigma2=10, **a)
        try:
            ref.processing_single_scene(args)
            raise SystemExit("the reference converted images without cv2?")
        except AttributeError as e:  # the first cv2 call: everything that is pinned has been written
            assert "cv2" in str(e), e
        shutil.rmtree(os.path.join(save, "images"), ignore_errors=True)
        shutil.rmtree(os.path.join(dense, "images"), ignore_errors=True)
        with open(os.path.join(case, "args.txt"), "w") as f:
            f.write(" ".join("--%s %s" % (k, v) for k, v in sorted(a.items())) + "\n")
        print(name, sorted(os.listdir(save)), len(os.listdir(os.path.join(save, "cams"))), "cams")


if __name__ == "__main__":
    main()
