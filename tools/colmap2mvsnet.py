#!/usr/bin/env python
"""COLMAP sparse model -> the dense folder the PatchMatch path reads (cams/, images/, pair.txt).

Host-side data preparation of SURVEY.md section 8 (f4); same command line and same outputs as the reference's
converter (`colmap2mvsnet.py:304-473`, arguments `:476-494`), written against numpy 2 / PIL (no OpenCV in
this image).  cams/ and pair.txt are pinned byte for byte against files the reference's own script wrote in the build
container (tests/golden/colmap, made by tests/golden/make_colmap_golden.py: text and binary model, --max_d 0,
--scale_factor 2); the image conversion at its end needs cv2 and is tested as documented behaviour only
(tests/test_colmap_converter.py).

What is kept, because the C++ side depends on it:
  * images are re-indexed 0..n-1 in ascending COLMAP image id (`:354-357`); file names `%08d_cam.txt`, `%08d.jpg`
  * cams file: "extrinsic" + 4x4 world->camera rows, "intrinsic" + 3x3, then `depth_min interval depth_num depth_max`
    printed with `%f` (`:436-449`); the matrices are printed with Python's shortest round-trip float repr
  * intrinsics: fx fy cx cy of the COLMAP camera divided by --scale_factor, distortion ignored (`:339-351`)
  * depth range: z of the image's triangulated points in its camera frame, 1 % / 99 % order statistics relaxed by
    0.75 / 1.25 (`:370-384`); depth_num = --max_d, or the inverse-depth count when --max_d 0 (`:387-400`)
  * pair score = number of observations of view i whose 3-D point is also seen by view j, zero when the 75 % order
    statistic of the triangulation angles is below 1 degree (`:280-302`; the Gaussian weighting is commented out in
    the reference, --theta0/--sigma1/--sigma2 are accepted and unused); up to 20 partners by descending score (`:424-428`)
  * pair.txt: count, then per view "id" and "n id score id score ..." with integer scores (`:450-456`)
  * images: zero-padded (bottom/right) to the largest width/height, nearest-neighbour resize by --scale_factor,
    written as JPEG (`:458-474`)

What is different: the model is parsed into flat arrays instead of per-record tuples and the O(n^2) view scoring is a
sorted-set intersection per pair instead of a list scan per observation (ETH3D scenes: seconds instead of minutes).
"""
import argparse
import os
import shutil
import struct

import numpy as np

# COLMAP camera models: id -> (name, number of parameters, index of fx, fy, cx, cy in the parameter vector)
_MODELS = {
    0: ("SIMPLE_PINHOLE", 3, (0, 0, 1, 2)),
    1: ("PINHOLE", 4, (0, 1, 2, 3)),
    2: ("SIMPLE_RADIAL", 4, (0, 0, 1, 2)),
    3: ("RADIAL", 5, (0, 0, 1, 2)),
    4: ("OPENCV", 8, (0, 1, 2, 3)),
    5: ("OPENCV_FISHEYE", 8, (0, 1, 2, 3)),
    6: ("FULL_OPENCV", 12, (0, 1, 2, 3)),
    7: ("FOV", 5, (0, 1, 2, 3)),
    8: ("SIMPLE_RADIAL_FISHEYE", 4, (0, 0, 1, 2)),
    9: ("RADIAL_FISHEYE", 5, (0, 0, 1, 2)),
    10: ("THIN_PRISM_FISHEYE", 12, (0, 1, 2, 3)),
}
_MODEL_BY_NAME = {name: (n, idx) for name, n, idx in _MODELS.values()}


class SparseModel:
    """cameras: {camera_id: (model name, width, height, params float64[])}
    views: list of dicts {id, qvec[4], tvec[3], camera_id, name, point3D_ids int64[]} in file order
    points: (ids int64[n] sorted ascending, xyz float64[n,3] in the same order)"""

    def __init__(self, cameras, views, point_ids, point_xyz):
        self.cameras = cameras
        self.views = views
        order = np.argsort(point_ids, kind="stable")
        self.point_ids = np.asarray(point_ids, np.int64)[order]
        self.point_xyz = np.asarray(point_xyz, np.float64).reshape(-1, 3)[order]

    def xyz_of(self, ids):
        """Coordinates of the given point ids (all must exist, like the reference's dict lookup)."""
        pos = np.searchsorted(self.point_ids, ids)
        if np.any(pos >= len(self.point_ids)) or np.any(self.point_ids[np.minimum(pos, len(self.point_ids) - 1)] != ids):
            raise KeyError("images file refers to a 3-D point that points3D does not hold")
        return self.point_xyz[pos]


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_model_text(folder):
    cameras = {}
    for line in _data_lines(os.path.join(folder, "cameras.txt")):
        e = line.split()
        cameras[int(e[0])] = (e[1], int(e[2]), int(e[3]), np.array(e[4:], np.float64))
    views = []
    # images.txt: two lines per image; the second (2-D points) may be empty, so blank lines cannot be skipped there
    with open(os.path.join(folder, "images.txt"), "r") as f:
        while True:
            line = f.readline()
            if not line:
                break
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            e = line.split()
            obs = f.readline().split()
            views.append(dict(id=int(e[0]), qvec=np.array(e[1:5], np.float64), tvec=np.array(e[5:8], np.float64),
                              camera_id=int(e[8]), name=e[9], point3D_ids=np.array(obs[2::3], np.int64)))
    ids, xyz = [], []
    for line in _data_lines(os.path.join(folder, "points3D.txt")):
        e = line.split(None, 4)
        ids.append(int(e[0]))
        xyz.append((float(e[1]), float(e[2]), float(e[3])))
    return SparseModel(cameras, views, np.array(ids, np.int64), np.array(xyz, np.float64))


def read_model_binary(folder):
    cameras = {}
    with open(os.path.join(folder, "cameras.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cam_id, model_id, w, h = struct.unpack("<iiQQ", f.read(24))
            name, n_par, _ = _MODELS[model_id]
            cameras[cam_id] = (name, w, h, np.frombuffer(f.read(8 * n_par), "<f8").copy())
    views = []
    with open(os.path.join(folder, "images.bin"), "rb") as f:
        buf = f.read()
    (n,) = struct.unpack_from("<Q", buf, 0)
    off = 8
    for _ in range(n):
        head = struct.unpack_from("<idddddddi", buf, off)
        off += 64
        end = buf.index(b"\x00", off)
        name = buf[off:end].decode("utf-8")
        off = end + 1
        (n_obs,) = struct.unpack_from("<Q", buf, off)
        off += 8
        obs = np.frombuffer(buf, np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]), n_obs, off)
        off += 24 * n_obs
        views.append(dict(id=head[0], qvec=np.array(head[1:5]), tvec=np.array(head[5:8]), camera_id=head[8],
                          name=name, point3D_ids=obs["id"].astype(np.int64)))
    with open(os.path.join(folder, "points3D.bin"), "rb") as f:
        buf = f.read()
    (n,) = struct.unpack_from("<Q", buf, 0)
    off = 8
    ids = np.empty(n, np.int64)
    xyz = np.empty((n, 3), np.float64)
    for k in range(n):
        pid, x, y, z = struct.unpack_from("<Qddd", buf, off)
        (track,) = struct.unpack_from("<Q", buf, off + 43)
        off += 43 + 8 + 8 * track
        ids[k] = pid
        xyz[k] = (x, y, z)
    return SparseModel(cameras, views, ids, xyz)


def read_model(folder, ext):
    return read_model_text(folder) if ext == ".txt" else read_model_binary(folder)


def rotation_of(q):
    """COLMAP quaternion (w, x, y, z) -> 3x3 rotation; term order of `colmap2mvsnet.py:251-262`."""
    w, x, y, z = q
    return np.array([
        [1 - 2 * y**2 - 2 * z**2, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x**2 - 2 * z**2, 2 * y * z - 2 * w * x],
        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x**2 - 2 * y**2]])


def intrinsic_of(camera, scale_factor):
    model, _, _, par = camera
    _, (ifx, ify, icx, icy) = _MODEL_BY_NAME[model]
    return np.array([[par[ifx] / scale_factor, 0, par[icx] / scale_factor],
                     [0, par[ify] / scale_factor, par[icy] / scale_factor],
                     [0, 0, 1]])


def depth_range_of(z, K, E, max_d, interval_scale):
    """`colmap2mvsnet.py:370-402` for one view; z = camera-frame depths of its triangulated points."""
    depth_min = depth_max = 0
    if len(z):
        zs = np.sort(z)
        depth_min = zs[int(len(zs) * .01)] * 0.75
        depth_max = zs[int(len(zs) * .99)] * 1.25
    if max_d == 0:
        # number of inverse-depth samples whose first step moves the principal point's ray by one pixel
        R, t = E[:3, :3], E[:3, 3]
        Kinv, Rinv = np.linalg.inv(K), np.linalg.inv(R)
        P1 = Rinv @ (Kinv @ [K[0, 2], K[1, 2], 1] * depth_min - t)
        P2 = Rinv @ (Kinv @ [K[0, 2] + 1, K[1, 2], 1] * depth_min - t)
        depth_num = (1 / depth_min - 1 / depth_max) / (1 / depth_min - 1 / (depth_min + np.linalg.norm(P2 - P1)))
    else:
        depth_num = max_d
    return depth_min, (depth_max - depth_min) / (depth_num - 1) / interval_scale, depth_num, depth_max


def pair_score(ids_i, ids_j, centre_i, centre_j, model):
    """`calc_score`, `colmap2mvsnet.py:280-302`: every observation of view i (a repeated id counts every time it
    appears) whose point view j also observes scores 1; all or nothing on the 75 % triangulation angle."""
    shared = ids_i[(ids_i != -1) & np.isin(ids_i, ids_j)]
    if len(shared) == 0:
        return 0.0
    p = model.xyz_of(shared)
    a, b = centre_i - p, centre_j - p
    cosine = np.einsum("ij,ij->i", a, b) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
    with np.errstate(invalid="ignore"):
        theta = np.sort((180 / np.pi) * np.arccos(cosine))
    # a NaN angle (|cos| rounded above 1) sorts last here; Python's sorted() in the reference leaves the order
    # of a list with NaNs unspecified, so only NaN-free inputs are comparable
    if theta[int(len(theta) * 0.75)] < 1:
        return 0.0
    return float(len(shared))


def convert(dense_folder, save_folder, max_d=192, interval_scale=1.0, scale_factor=1.0, model_ext=".txt",
            model_subdir="dslr_calibration_undistorted", write_images=True, verbose=True):
    """Returns (intrinsics, extrinsics, depth_ranges, view_sel) as lists indexed by the new 0-based view index."""
    model = read_model(os.path.join(dense_folder, model_subdir), model_ext)
    views = sorted(model.views, key=lambda v: v["id"])
    n = len(views)
    cam_dir = os.path.join(save_folder, "cams")
    img_dir = os.path.join(save_folder, "images")
    os.makedirs(save_folder, exist_ok=True)
    for d in (img_dir, cam_dir):
        if os.path.exists(d):
            shutil.rmtree(d)
    os.makedirs(img_dir)
    os.makedirs(cam_dir)

    K = [intrinsic_of(model.cameras[v["camera_id"]], scale_factor) for v in views]
    E = []
    for v in views:
        e = np.zeros((4, 4))
        e[:3, :3] = rotation_of(v["qvec"])
        e[:3, 3] = v["tvec"]
        e[3, 3] = 1
        E.append(e)

    ranges = []
    for v, k, e in zip(views, K, E):
        ids = v["point3D_ids"]
        p = model.xyz_of(ids[ids != -1])
        z = (np.concatenate([p, np.ones((len(p), 1))], axis=1) @ e[2]) if len(p) else np.zeros(0)
        ranges.append(depth_range_of(z, k, e, max_d, interval_scale))

    centres = [-(e[:3, :3].T @ e[:3, 3]) for e in E]
    score = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            score[i, j] = score[j, i] = pair_score(views[i]["point3D_ids"], views[j]["point3D_ids"],
                                                   centres[i], centres[j], model)
    num_view = min(20, n - 1)
    view_sel = [[(int(k), score[i, k]) for k in np.argsort(score[i])[::-1][:num_view]] for i in range(n)]

    for i in range(n):
        with open(os.path.join(cam_dir, "%08d_cam.txt" % i), "w") as f:
            f.write("extrinsic\n")
            for row in E[i]:
                f.write("".join(str(float(x)) + " " for x in row) + "\n")
            f.write("\nintrinsic\n")
            for row in K[i]:
                f.write("".join(str(float(x)) + " " for x in row) + "\n")
            f.write("\n%f %f %f %f\n" % ranges[i])
    with open(os.path.join(save_folder, "pair.txt"), "w") as f:
        f.write("%d\n" % n)
        for i, sel in enumerate(view_sel):
            f.write("%d\n%d " % (i, len(sel)))
            for k, s in sel:
                f.write("%d %d " % (k, s))
            f.write("\n")

    if write_images:
        _convert_images(os.path.join(dense_folder, "images"), [v["name"] for v in views], img_dir, scale_factor)
    if verbose:
        print("converted %d views -> %s" % (n, save_folder))
    return K, E, ranges, view_sel


def nearest_resize(img, new_w, new_h):
    """cv::resize(..., INTER_NEAREST): source index = min(floor(dst * src/dst), src - 1) per axis."""
    h, w = img.shape[:2]
    sx = np.minimum(np.floor(np.arange(new_w) * (w / new_w)).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(new_h) * (h / new_h)).astype(np.int64), h - 1)
    return img[sy][:, sx]


def _convert_images(src_dir, names, dst_dir, scale_factor):
    from PIL import Image

    sizes = []
    for name in names:
        with Image.open(os.path.join(src_dir, name)) as im:
            sizes.append(im.size)
    max_w = max(s[0] for s in sizes)
    max_h = max(s[1] for s in sizes)
    for i, name in enumerate(names):
        with Image.open(os.path.join(src_dir, name)) as im:
            img = np.asarray(im.convert("RGB"))          # cv::imread default: 3 channels, alpha dropped
        pad = np.zeros((max_h, max_w, 3), np.uint8)
        pad[:img.shape[0], :img.shape[1]] = img
        out = nearest_resize(pad, int(max_w / scale_factor), int(max_h / scale_factor))
        # cv::imwrite's JPEG defaults: quality 95, 4:2:0 chroma
        Image.fromarray(out).save(os.path.join(dst_dir, "%08d.jpg" % i), quality=95, subsampling=2)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Convert colmap camera")
    ap.add_argument("--dense_folder", required=True, type=str)
    ap.add_argument("--save_folder", required=True, type=str)
    ap.add_argument("--max_d", type=int, default=192)
    ap.add_argument("--interval_scale", type=float, default=1)
    ap.add_argument("--scale_factor", type=float, default=1)
    ap.add_argument("--theta0", type=float, default=5)
    ap.add_argument("--sigma1", type=float, default=1)
    ap.add_argument("--sigma2", type=float, default=10)
    ap.add_argument("--model_ext", type=str, default=".txt", choices=[".txt", ".bin"])
    ap.add_argument("--model_subdir", type=str, default="dslr_calibration_undistorted",
                    help="folder of the sparse model inside --dense_folder (the reference hard-codes ETH3D's)")
    a = ap.parse_args(argv)
    convert(a.dense_folder, a.save_folder, a.max_d, a.interval_scale, a.scale_factor, a.model_ext, a.model_subdir)


if __name__ == "__main__":
    main()
