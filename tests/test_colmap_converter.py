"""tools/colmap2mvsnet.py: COLMAP sparse model -> cams/, images/, pair.txt as the reference's
converter lays them out (`colmap2mvsnet.py:304-473`), read back through the drop-in's own readers.  No GPU needed."""
import ctypes as C
import importlib.util
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "apd-mvs_amd", "_build", "libapd_host.so")

_spec = importlib.util.spec_from_file_location("colmap2mvsnet", os.path.join(ROOT, "tools", "colmap2mvsnet.py"))
conv = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(conv)


def _quat(R):
    """Rotation matrix -> COLMAP (w, x, y, z); rotations here have a comfortably positive trace."""
    w = np.sqrt(1 + np.trace(R)) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def _scene(seed=0):
    """Five views: four on an arc around a point cloud at z ~ 4..6, the fifth 1 mm beside the first (triangulation
    angles far below 1 degree).  COLMAP ids are shuffled and non-contiguous on purpose."""
    rng = np.random.RandomState(seed)
    pts = np.column_stack([rng.uniform(-1, 1, 400), rng.uniform(-1, 1, 400), rng.uniform(4, 6, 400)])
    point_ids = rng.permutation(np.arange(10, 10 + 3 * 400, 3))[:400]
    cams = {3: ("PINHOLE", 640, 480, [600.0, 610.0, 320.0, 240.0]),
            8: ("SIMPLE_RADIAL", 600, 440, [500.0, 300.0, 220.0, 0.01])}
    views = []
    image_ids = [7, 2, 19, 11, 30]
    for k, a in enumerate([0.0, 0.15, 0.3, -0.2, 0.0]):
        R = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
        centre = np.array([5 * np.sin(a), 0.05 * k, 5 - 5 * np.cos(a)]) if k != 4 else np.array([1e-3, 0, 0])
        t = -R @ centre
        seen = rng.rand(400) < (0.6 if k != 3 else 0.3)
        obs_ids = np.where(seen, point_ids, -1)[rng.permutation(400)][:300]
        views.append(dict(id=image_ids[k], q=_quat(R), t=t, R=R, cam=3 if k % 2 == 0 else 8,
                          name="dslr/im%d.png" % k, ids=obs_ids))
    return cams, views, point_ids, pts


def _write_text(folder, cams, views, point_ids, pts):
    os.makedirs(folder)
    with open(os.path.join(folder, "cameras.txt"), "w") as f:
        f.write("# Camera list\n")
        for cid, (m, w, h, p) in cams.items():
            f.write("%d %s %d %d %s\n" % (cid, m, w, h, " ".join(repr(x) for x in p)))
    with open(os.path.join(folder, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image\n")
        for v in views:
            f.write("%d %s %s %d %s\n" % (v["id"], " ".join(repr(float(x)) for x in v["q"]),
                                          " ".join(repr(float(x)) for x in v["t"]), v["cam"], v["name"]))
            f.write(" ".join("%.2f %.2f %d" % (1.5 * i, 2.5 * i, pid) for i, pid in enumerate(v["ids"])) + "\n")
    with open(os.path.join(folder, "points3D.txt"), "w") as f:
        f.write("# 3D point list\n")
        for pid, p in zip(point_ids, pts):
            f.write("%d %r %r %r 10 20 30 0.5 1 2 3 4\n" % (pid, float(p[0]), float(p[1]), float(p[2])))


def _write_binary(folder, cams, views, point_ids, pts):
    os.makedirs(folder)
    model_id = {"PINHOLE": 1, "SIMPLE_RADIAL": 2}
    with open(os.path.join(folder, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, (m, w, h, p) in cams.items():
            f.write(struct.pack("<iiQQ", cid, model_id[m], w, h) + struct.pack("<%dd" % len(p), *p))
    with open(os.path.join(folder, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(views)))
        for v in views:
            f.write(struct.pack("<idddddddi", v["id"], *v["q"], *v["t"], v["cam"]) + v["name"].encode() + b"\0")
            f.write(struct.pack("<Q", len(v["ids"])))
            for i, pid in enumerate(v["ids"]):
                f.write(struct.pack("<ddq", 1.5 * i, 2.5 * i, int(pid)))
    with open(os.path.join(folder, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(pts)))
        for pid, p in zip(point_ids, pts):
            f.write(struct.pack("<QdddBBBd", int(pid), p[0], p[1], p[2], 10, 20, 30, 0.5))
            f.write(struct.pack("<Q", 2) + struct.pack("<iiii", 1, 2, 3, 4))


def _read_pairs(path):
    tok = open(path).read().split()
    n = int(tok[0])
    pos, out = 1, {}
    for _ in range(n):
        ref, m = int(tok[pos]), int(tok[pos + 1])
        out[ref] = [(int(tok[pos + 2 + 2 * k]), int(tok[pos + 3 + 2 * k])) for k in range(m)]
        pos += 2 + 2 * m
    assert pos == len(tok)
    return out


@pytest.fixture(scope="module")
def converted(tmp_path_factory):
    from PIL import Image

    root = tmp_path_factory.mktemp("colmap")
    cams, views, point_ids, pts = _scene()
    dense = root / "scene"
    _write_text(str(dense / "dslr_calibration_undistorted"), cams, views, point_ids, pts)
    _write_binary(str(dense / "sparse_bin"), cams, views, point_ids, pts)
    os.makedirs(dense / "images" / "dslr")
    rng = np.random.RandomState(1)
    sizes = [(64, 48), (60, 44), (64, 48), (60, 44), (64, 40)]
    raw = []
    for k, (w, h) in enumerate(sizes):
        img = np.clip(np.kron(rng.randint(40, 215, (h // 4, w // 4, 3)), np.ones((4, 4, 1))), 0, 255).astype(np.uint8)
        raw.append(img)
        Image.fromarray(img).save(str(dense / "images" / "dslr" / ("im%d.png" % k)))
    out_txt = conv.convert(str(dense), str(root / "out_txt"), verbose=False)
    out_bin = conv.convert(str(dense), str(root / "out_bin"), model_ext=".bin", model_subdir="sparse_bin", verbose=False)
    return dict(root=root, views=views, cams=cams, point_ids=point_ids, pts=pts, raw=raw, sizes=sizes,
                out_txt=out_txt, out_bin=out_bin)


def _order(views):
    return sorted(range(len(views)), key=lambda k: views[k]["id"])


def test_text_and_binary_models_give_the_same_files(converted):
    root = converted["root"]
    names = ["pair.txt"] + ["cams/%08d_cam.txt" % i for i in range(5)] + ["images/%08d.jpg" % i for i in range(5)]
    for n in names:
        assert (root / "out_txt" / n).read_bytes() == (root / "out_bin" / n).read_bytes(), n


def test_cams_are_read_back_by_the_dropin_reader(converted, pkg):
    """Views re-indexed by ascending COLMAP id; [R t], K (distortion dropped, f -> fx = fy) and the 1 % / 99 % depth
    order statistics relaxed by 0.75 / 1.25 arrive in `ReadCamera` (APD.cpp:51-92)."""
    assert os.path.exists(HOST_LIB), "run __graft_entry__.build() first"
    pkg.lib()
    L = C.CDLL(HOST_LIB)
    L.apdhost_read_camera.argtypes = [C.c_char_p, C.c_void_p]
    views, cams = converted["views"], converted["cams"]
    xyz = dict(zip(converted["point_ids"], converted["pts"]))
    for new_idx, k in enumerate(_order(views)):
        v = views[k]
        cam = pkg.Camera()
        path = str(converted["root"] / "out_txt" / "cams" / ("%08d_cam.txt" % new_idx))
        assert L.apdhost_read_camera(path.encode(), C.byref(cam)) == 0
        assert np.allclose(np.array(list(cam.R)).reshape(3, 3), v["R"], atol=1e-6)
        assert np.allclose(list(cam.t), v["t"], atol=1e-5)
        p = cams[v["cam"]][3]
        K = [p[0], 0, p[2], 0, p[1], p[3], 0, 0, 1] if v["cam"] == 3 else [p[0], 0, p[1], 0, p[0], p[2], 0, 0, 1]
        assert np.allclose(list(cam.K), K, atol=1e-4)
        z = np.sort([(v["R"] @ xyz[i] + v["t"])[2] for i in v["ids"] if i != -1])
        assert abs(cam.depth_min - z[int(len(z) * .01)] * 0.75) < 1e-5
        assert abs(cam.depth_max - z[int(len(z) * .99)] * 1.25) < 1e-5
        last = open(path).read().split()[-4:]
        assert float(last[2]) == 192.0
        assert abs(float(last[1]) - (float(last[3]) - float(last[0])) / 191) < 1e-5


def test_pair_scores_count_shared_points_and_drop_degenerate_baselines(converted):
    views = converted["views"]
    order = _order(views)
    pairs = _read_pairs(str(converted["root"] / "out_txt" / "pair.txt"))
    assert sorted(pairs) == list(range(5))
    new_of = {k: i for i, k in enumerate(order)}
    for k in range(5):
        got = dict(pairs[new_of[k]])
        assert len(got) == 4                                   # min(20, n - 1) partners, self included only as filler
        for j in range(5):
            if j == k:
                continue
            a = views[k]["ids"]
            shared = int(np.sum((a != -1) & np.isin(a, views[j]["ids"])))
            degenerate = {k, j} == {0, 4}                      # 1 mm baseline: every angle < 1 degree
            if new_of[j] in got:
                assert got[new_of[j]] == (0 if degenerate else shared), (k, j)
        scores = [s for _, s in pairs[new_of[k]]]
        assert scores == sorted(scores, reverse=True)


def test_images_are_padded_to_the_largest_and_written_as_jpeg(converted, pkg):
    from PIL import Image

    views, raw = converted["views"], converted["raw"]
    for new_idx, k in enumerate(_order(views)):
        im = np.asarray(Image.open(str(converted["root"] / "out_txt" / "images" / ("%08d.jpg" % new_idx))).convert("RGB"))
        assert im.shape == (48, 64, 3)
        h, w = raw[k].shape[:2]
        # 4x4 flat blocks survive quality-95 JPEG within a few grey levels away from the block edges
        assert np.abs(im[1:h:4, 1:w:4].astype(int) - raw[k][1:h:4, 1:w:4].astype(int)).mean() < 6
        assert im[h + 4:, :].mean() < 8 if h + 4 < 48 else True
        assert im[:, w + 4:].mean() < 8 if w + 4 < 64 else True


def test_scale_factor_and_inverse_depth_count(converted, tmp_path):
    """--scale_factor divides the intrinsics and the image size (nearest neighbour); --max_d 0 derives the number of
    depth samples from a one-pixel step at depth_min (`colmap2mvsnet.py:387-400`)."""
    from PIL import Image

    dense = str(converted["root"] / "scene")
    K, E, ranges, _ = conv.convert(dense, str(tmp_path / "half"), max_d=0, scale_factor=2.0, verbose=False)
    K1 = converted["out_txt"][0]
    assert np.allclose(K[0][:2], K1[0][:2] / 2)
    assert Image.open(str(tmp_path / "half" / "images" / "00000000.jpg")).size == (32, 24)
    dmin, interval, num, dmax = ranges[0]
    step = dmin / K[0][0, 0]                                   # a pixel at depth_min, in world units (fx pixels per unit)
    expect = (1 / dmin - 1 / dmax) / (1 / dmin - 1 / (dmin + step))
    assert abs(num - expect) / expect < 1e-9
    assert abs(interval - (dmax - dmin) / (num - 1)) < 1e-12


def test_nearest_resize_picks_floor_of_scaled_index():
    a = np.arange(7 * 5).reshape(5, 7)
    out = conv.nearest_resize(a, 3, 2)
    assert np.array_equal(out, a[[0, 2]][:, [0, 2, 4]])


def test_quaternion_convention():
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])   # 90 degrees about z
    assert np.allclose(conv.rotation_of([np.sqrt(0.5), 0, 0, np.sqrt(0.5)]), R)


# ---- pinned against the reference's own converter ------------------------------------------------------------------------
# tests/golden/colmap/<case>/ holds a synthetic COLMAP model (input/) and the cams/ + pair.txt that the REFERENCE script wrote
# for it in the build container (expected/; tests/golden/make_colmap_golden.py explains how it was run there without cv2).
# The image conversion at the end of the reference script needs cv2 and is not part of the vectors.
_GOLDEN = os.path.join(ROOT, "tests", "golden", "colmap")
_CASES = {
    "text_default": dict(model_ext=".txt", max_d=192, interval_scale=1, scale_factor=1),
    "binary_inverse_depth_scale2": dict(model_ext=".bin", max_d=0, interval_scale=0.8, scale_factor=2),
}


@pytest.mark.parametrize("case", sorted(_CASES))
def test_cams_and_pairs_equal_the_reference_converter_byte_for_byte(case, tmp_path):
    src = os.path.join(_GOLDEN, case)
    assert open(os.path.join(src, "args.txt")).read().split() == [t for k, v in sorted(_CASES[case].items()) for t in ("--" + k, str(v))]
    out = str(tmp_path / "out")
    conv.convert(os.path.join(src, "input"), out, write_images=False, verbose=False, **_CASES[case])
    expected = os.path.join(src, "expected")
    names = sorted(os.listdir(os.path.join(expected, "cams")))
    assert names == sorted(os.listdir(os.path.join(out, "cams"))) and len(names) == 5
    for n in names:
        want = open(os.path.join(expected, "cams", n)).read()
        got = open(os.path.join(out, "cams", n)).read()
        assert got == want, "%s differs from the reference's file:\n%s\n--- reference ---\n%s" % (n, got, want)
    assert open(os.path.join(out, "pair.txt")).read() == open(os.path.join(expected, "pair.txt")).read()
