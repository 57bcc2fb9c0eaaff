#!/usr/bin/env python3
"""Writes a synthetic dense folder in the reference's layout (pair.txt, cams/%08d_cam.txt, images/%08d.pgm|jpg) from the
analytic scene generator: `num_views` cameras on a ring looking at slanted, textured planes.

  python tools/make_synthetic_dense.py <folder> --width 1920 --height 1080 --views 16 --src 10 [--textureless 0.2] [--jpeg] [--hard] [--gt]

--hard: the synth.HARD scene (slabs in front of the planes, per-view gain / offset, sources aiming off the target).
--gt:   also writes gt/%08d.npy (+ _normal.npy, _textured.npy), the analytic z-depth, world-frame normal and has-texture mask of every view (tools/jacobi_vs_gs.py compares both orders of views against it).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_dense_folder(folder, synth, width, height, num_views, num_src, seed=0, textureless=0.0, jpeg=False, device="cpu", hard=False, gt=False):
    os.makedirs(os.path.join(folder, "images"), exist_ok=True)
    os.makedirs(os.path.join(folder, "cams"), exist_ok=True)
    sc = synth.make_scene(width, height, num_views - 1, seed=seed, textureless=textureless, device=device, keep_view_depths=gt,
                          **(synth.HARD if hard else {}))
    imgs = sc.images_numpy()
    if gt:
        os.makedirs(os.path.join(folder, "gt"), exist_ok=True)
        for i in range(num_views):
            np.save(os.path.join(folder, "gt", "%08d.npy" % i), sc.view_depths[i].detach().cpu().numpy())
            np.save(os.path.join(folder, "gt", "%08d_normal.npy" % i), sc.view_normals[i].detach().cpu().numpy())
            np.save(os.path.join(folder, "gt", "%08d_textured.npy" % i), sc.view_textured[i].detach().cpu().numpy())
    for i in range(num_views):
        a = imgs[i].astype(np.uint8)
        if jpeg:
            from PIL import Image
            Image.fromarray(a, "L").save(os.path.join(folder, "images", "%08d.jpg" % i), quality=95)
        else:
            with open(os.path.join(folder, "images", "%08d.pgm" % i), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (width, height) + a.tobytes())
        R, t, K = sc.R[i].reshape(3, 3), sc.t[i], sc.K[i].reshape(3, 3)
        txt = "extrinsic\n"
        for r in range(3):
            txt += "%.9g %.9g %.9g %.9g\n" % (R[r, 0], R[r, 1], R[r, 2], t[r])
        txt += "0 0 0 1\n\nintrinsic\n"
        for r in range(3):
            txt += "%.9g %.9g %.9g\n" % (K[r, 0], K[r, 1], K[r, 2])
        txt += "\n%.9g 0.01 192 %.9g\n" % (sc.depth_min, sc.depth_max)
        with open(os.path.join(folder, "cams", "%08d_cam.txt" % i), "w") as f:
            f.write(txt)
    pair = "%d\n" % num_views
    for i in range(num_views):
        srcs = sorted((j for j in range(num_views) if j != i), key=lambda j: (abs(j - i), j))[:num_src]
        pair += "%d\n%d %s\n" % (i, len(srcs), " ".join("%d %.1f" % (j, 100.0 - k) for k, j in enumerate(srcs)))
    with open(os.path.join(folder, "pair.txt"), "w") as f:
        f.write(pair)
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("folder")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--src", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--textureless", type=float, default=0.0)
    ap.add_argument("--jpeg", action="store_true")
    ap.add_argument("--hard", action="store_true")
    ap.add_argument("--gt", action="store_true")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.load_package()
    from apd_mvs_amd import synth
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    write_dense_folder(args.folder, synth, args.width, args.height, args.views, args.src, args.seed, args.textureless, args.jpeg, dev, hard=args.hard, gt=args.gt)
    print("wrote", args.folder)


if __name__ == "__main__":
    main()
