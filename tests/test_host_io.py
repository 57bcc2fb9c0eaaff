"""Host-side drop-in pieces (apd-mvs_amd/host): on-disk formats of the reference (.dmb/.bin, *_cam.txt),
grey JPEG input, resampling.  No GPU needed."""
import ctypes as C
import io
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "apd-mvs_amd", "_build", "libapd_host.so")


@pytest.fixture(scope="module")
def host(pkg):
    assert os.path.exists(HOST_LIB), "run __graft_entry__.build() first"
    pkg.lib()  # libapd_host.so depends on libapd_mi355x.so
    L = C.CDLL(HOST_LIB)
    L.apdhost_format_index.restype = C.c_char_p
    L.apdhost_format_index.argtypes = [C.c_int]
    ip = C.POINTER(C.c_int)
    fp = C.POINTER(C.c_float)
    L.apdhost_read_bin_mat.argtypes = [C.c_char_p, ip, ip, ip, C.c_void_p, C.c_size_t]
    L.apdhost_write_bin_mat.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.apdhost_read_camera.argtypes = [C.c_char_p, C.c_void_p]
    L.apdhost_read_gray_image.argtypes = [C.c_char_p, ip, ip, fp, C.c_size_t]
    L.apdhost_resize_linear.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    L.apdhost_rescale_nearest_f32.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    return L


def test_format_index(host):
    assert host.apdhost_format_index(7) == b"00000007"
    assert host.apdhost_format_index(12345678) == b"12345678"


@pytest.mark.parametrize("dtype,code,ch", [(np.float32, 5, 1), (np.float32, 21, 3), (np.uint8, 0, 1), (np.int32, 4, 1)])
def test_bin_mat_layout_and_round_trip(host, tmp_path, dtype, code, ch):
    """int32 version=1, rows, cols, OpenCV type code, then the raw row-major payload (APD.cpp:3-49)."""
    rows, cols = 5, 7
    rng = np.random.RandomState(0)
    a = (rng.rand(rows, cols * ch) * 200).astype(dtype)
    p = str(tmp_path / "m.dmb").encode()
    assert host.apdhost_write_bin_mat(p, rows, cols, code, a.ctypes.data) == 0
    raw = open(p, "rb").read()
    assert struct.unpack("<4i", raw[:16]) == (1, rows, cols, code)
    assert raw[16:] == a.tobytes()
    r, c, t = C.c_int(), C.c_int(), C.c_int()
    out = np.zeros_like(a)
    assert host.apdhost_read_bin_mat(p, C.byref(r), C.byref(c), C.byref(t), out.ctypes.data, out.nbytes) == 0
    assert (r.value, c.value, t.value) == (rows, cols, code)
    assert np.array_equal(out, a)


def test_bin_mat_rejects_wrong_version_and_missing_file(host, tmp_path):
    p = tmp_path / "bad.dmb"
    p.write_bytes(struct.pack("<4i", 2, 1, 1, 5) + b"\0\0\0\0")
    r, c, t = C.c_int(), C.c_int(), C.c_int()
    assert host.apdhost_read_bin_mat(str(p).encode(), C.byref(r), C.byref(c), C.byref(t), None, 0) != 0
    assert host.apdhost_read_bin_mat(str(tmp_path / "nope.dmb").encode(), C.byref(r), C.byref(c), C.byref(t), None, 0) != 0


def test_read_camera(host, pkg, tmp_path):
    """MVSNet-style cam file: extrinsic [R t] rows, intrinsic K, `depth_min interval depth_num depth_max`;
    camera centre c = -R^T t (APD.cpp:51-92)."""
    R = np.array([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])
    t = np.array([0.1, -0.2, 0.3])
    txt = "extrinsic\n"
    for i in range(3):
        txt += "%f %f %f %f\n" % (R[i, 0], R[i, 1], R[i, 2], t[i])
    txt += "0.0 0.0 0.0 1.0\n\nintrinsic\n1000.5 0 320.25\n0 1001.5 240.75\n0 0 1\n\n0.5 0.01 192 7.25\n"
    p = tmp_path / "00000003_cam.txt"
    p.write_text(txt)
    cam = pkg.Camera()
    assert host.apdhost_read_camera(str(p).encode(), C.byref(cam)) == 0
    assert np.allclose(np.array(list(cam.R)).reshape(3, 3), R, atol=1e-6)
    assert np.allclose(list(cam.t), t, atol=1e-6)
    assert np.allclose(list(cam.K), [1000.5, 0, 320.25, 0, 1001.5, 240.75, 0, 0, 1], atol=1e-4)
    assert np.allclose(list(cam.c), -R.T @ t, atol=1e-6)
    assert abs(cam.depth_min - 0.5) < 1e-7 and abs(cam.depth_max - 7.25) < 1e-7


def _read_image(host, stem, shape):
    r, c = C.c_int(), C.c_int()
    out = np.zeros(shape, np.float32)
    rc = host.apdhost_read_gray_image(str(stem).encode(), C.byref(r), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    return rc, (r.value, c.value), out


def test_pgm_input(host, tmp_path):
    a = (np.arange(6 * 9) % 256).astype(np.uint8).reshape(6, 9)
    (tmp_path / "00000001.pgm").write_bytes(b"P5\n# comment\n9 6\n255\n" + a.tobytes())
    rc, shp, out = _read_image(host, tmp_path / "00000001", (6, 9))
    assert rc == 0 and shp == (6, 9)
    assert np.array_equal(out, a.astype(np.float32))


@pytest.mark.parametrize("mode,subsampling,size", [("L", 0, (64, 48)), ("RGB", 0, (61, 45)), ("RGB", 2, (70, 50)), ("RGB", 1, (33, 17))])
def test_baseline_jpeg_luma_matches_libjpeg(host, tmp_path, mode, subsampling, size):
    """cv::imread(IMREAD_GRAYSCALE) == libjpeg decoding straight to the Y plane.  PIL's draft('L') asks
    its bundled libjpeg for exactly that, so the two decoders must agree bit for bit."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(1)
    w, h = size
    base = rng.rand(h // 4 + 2, w // 4 + 2, 3)
    img = np.kron(base, np.ones((4, 4, 1)))[:h, :w] * 255
    img += rng.rand(h, w, 3) * 20
    img = np.clip(img, 0, 255).astype(np.uint8)
    im = PIL.fromarray(img if mode == "RGB" else img[..., 0], mode)
    path = tmp_path / "00000002.jpg"
    kw = {} if mode == "L" else {"subsampling": subsampling}
    im.save(path, quality=90, **kw)
    ref = PIL.open(path)
    ref.draft("L", ref.size)
    ref = np.asarray(ref.convert("L") if ref.mode != "L" else ref, np.float32)
    rc, shp, out = _read_image(host, tmp_path / "00000002", (h, w))
    assert rc == 0 and shp == (h, w)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("mode,subsampling,size,quality", [("L", 0, (64, 48), 90), ("L", 0, (37, 53), 60), ("RGB", 0, (61, 45), 92),
                                                           ("RGB", 2, (70, 50), 85), ("RGB", 1, (33, 17), 75), ("RGB", 2, (129, 130), 40)])
def test_progressive_jpeg_matches_libjpeg(host, tmp_path, mode, subsampling, size, quality):
    """cv::imread decodes progressive files (SOF2) like any other; so does the drop-in's decoder since round 2: spectral
    selection + successive approximation (T.81 Annex G: DC / AC first and refinement scans, EOB runs), then the same islow IDCT,
    chroma upsampling and colour conversion as for baseline files.  PIL's libjpeg is the golden decoder, grey and colour."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(size[0] * 7 + size[1] + quality)
    w, h = size
    base = rng.rand(h // 4 + 2, w // 4 + 2, 3)
    img = np.kron(base, np.ones((4, 4, 1)))[:h, :w] * 255
    img += rng.rand(h, w, 3) * 30
    img = np.clip(img, 0, 255).astype(np.uint8)
    path = tmp_path / "00000004.jpg"
    kw = {} if mode == "L" else {"subsampling": subsampling}
    Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(path, quality=quality, progressive=True, **kw)
    assert b"\xff\xc2" in path.read_bytes()[:1000]  # really a progressive file
    ref = Image.open(path)
    ref.draft("L", ref.size)
    ref = np.asarray(ref.convert("L") if ref.mode != "L" else ref, np.float32)
    rc, shp, out = _read_image(host, tmp_path / "00000004", (h, w))
    assert rc == 0 and shp == (h, w)
    assert np.array_equal(out, ref), int((out != ref).sum())
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    host.apdhost_read_color_image.argtypes = [C.c_char_p, ip, ip, fp, C.c_size_t]
    refc = np.asarray(Image.open(path).convert("RGB"))[..., ::-1].astype(np.float32)
    r, c = C.c_int(), C.c_int()
    outc = np.zeros((h, w, 3), np.float32)
    assert host.apdhost_read_color_image(str(tmp_path / "00000004").encode(), C.byref(r), C.byref(c), outc.ctypes.data_as(fp), outc.size) == 0
    assert np.array_equal(outc, refc), int((outc != refc).sum())


def test_progressive_jpeg_with_restart_markers(host, tmp_path):
    """Progressive + restart intervals (EOB runs and DC predictions reset at every RSTn)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(12)
    img = (np.kron(rng.rand(20, 28, 3), np.ones((4, 4, 1))) * 255).astype(np.uint8)
    path = tmp_path / "00000006.jpg"
    try:
        Image.fromarray(img, "RGB").save(path, quality=80, progressive=True, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("this PIL cannot write restart markers")
    ref = Image.open(path)
    ref.draft("L", ref.size)
    ref = np.asarray(ref.convert("L") if ref.mode != "L" else ref, np.float32)
    rc, shp, out = _read_image(host, tmp_path / "00000006", img.shape[:2])
    assert rc == 0 and np.array_equal(out, ref)


def test_resize_linear_power_of_two_is_centre_box(host):
    """cv::resize INTER_LINEAR at an exact 1/2 ratio averages the 2x2 block."""
    rng = np.random.RandomState(3)
    a = (rng.rand(12, 16) * 255).astype(np.float32)
    out = np.zeros((6, 8), np.float32)
    host.apdhost_resize_linear(a.ctypes.data_as(C.POINTER(C.c_float)), 12, 16, out.ctypes.data_as(C.POINTER(C.c_float)), 6, 8)
    ref = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]) / 4
    assert np.allclose(out, ref, atol=1e-4)
    # 1/4: the two middle rows/cols of each 4x4 cell with weights 0.5/0.5, not a full box
    out4 = np.zeros((3, 4), np.float32)
    host.apdhost_resize_linear(a.ctypes.data_as(C.POINTER(C.c_float)), 12, 16, out4.ctypes.data_as(C.POINTER(C.c_float)), 3, 4)
    ref4 = (a[1::4, 1::4] + a[1::4, 2::4] + a[2::4, 1::4] + a[2::4, 2::4]) / 4
    assert np.allclose(out4, ref4, atol=1e-4)


def test_rescale_nearest_keeps_swapped_scale_quirk(host):
    """RescaleMatToTargetSize divides the ROW by scale_x and the COLUMN by scale_y (APD.cpp:766-767)."""
    src = np.arange(5 * 7, dtype=np.float32).reshape(5, 7)
    new_rows, new_cols = 11, 13
    out = np.zeros((new_rows, new_cols), np.float32)
    host.apdhost_rescale_nearest_f32(src.ctypes.data_as(C.POINTER(C.c_float)), 5, 7, out.ctypes.data_as(C.POINTER(C.c_float)), new_rows, new_cols)
    sx, sy = np.float32(new_cols) / np.float32(7), np.float32(new_rows) / np.float32(5)
    ref = np.zeros_like(out)
    for r in range(new_rows):
        for c in range(new_cols):
            o_r, o_c = int(np.float32(r) / sx), int(np.float32(c) / sy)
            if 0 <= o_r < 5 and 0 <= o_c < 7:
                ref[r, c] = src[o_r, o_c]
    assert np.array_equal(out, ref)


def test_fusion_math_kernels(host):
    """acos / exp of the fusion arithmetic (contract C9, csrc/apd_fusion_math.h): fixed binary32 kernels shared by the
    host and the device fusion, within a few ulp of the correctly rounded function; NaN outside [-1, 1] as GetAngle expects."""
    fp = C.POINTER(C.c_float)
    host.apdhost_fusion_math.argtypes = [fp, C.c_int, C.c_int, fp]
    x = np.concatenate([np.linspace(-1, 1, 20001), [1.0, -1.0, 0.0, 0.5, -0.5, 0.9999999, 1e-9, -1e-9]]).astype(np.float32)
    out = np.zeros_like(x)
    host.apdhost_fusion_math(x.ctypes.data_as(fp), len(x), 0, out.ctypes.data_as(fp))
    ref = np.arccos(x.astype(np.float64))
    assert np.abs(out - ref).max() < 4e-7
    bad = np.array([1.0000001, -1.0000001, 2.0, np.nan], np.float32)
    o2 = np.zeros_like(bad)
    host.apdhost_fusion_math(bad.ctypes.data_as(fp), len(bad), 0, o2.ctypes.data_as(fp))
    assert np.isnan(o2).all()
    e = np.linspace(-30, 0, 10001).astype(np.float32)
    oe = np.zeros_like(e)
    host.apdhost_fusion_math(e.ctypes.data_as(fp), len(e), 1, oe.ctypes.data_as(fp))
    refe = np.exp(e.astype(np.float64))
    assert (np.abs(oe - refe) / refe).max() < 5e-7


@pytest.mark.parametrize("size", [(64, 48), (65, 47), (33, 17), (17, 9), (3, 5), (129, 130)])
def test_colour_jpeg_decode_matches_libjpeg(host, tmp_path, size):
    """ReadColorImage == cv::imread(IMREAD_COLOR) as far as libjpeg defines it (the fusion's point colours, APD.cpp:859):
    islow IDCT, fancy chroma upsampling (4:2:2, 4:2:0), fixed-point YCbCr -> RGB.  PIL decodes with libjpeg too, so its
    bytes are the golden vector."""
    Image = pytest.importorskip("PIL.Image")
    w, h = size
    rng = np.random.RandomState(w * 1000 + h)
    ys, xs = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xs * 0.2 + ys * 0.05), 127 + 100 * np.cos(ys * 0.3), (xs * 7 + ys * 3) % 256], -1)
    img = np.clip(img + rng.randn(h, w, 3) * 20, 0, 255).astype(np.uint8)
    ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
    host.apdhost_read_color_image.argtypes = [C.c_char_p, ip, ip, fp, C.c_size_t]
    for sub in (0, 1, 2):          # 4:4:4, 4:2:2, 4:2:0
        for q in (95, 60):
            stem = str(tmp_path / ("c_%d_%d" % (sub, q)))
            Image.fromarray(img, "RGB").save(stem + ".jpg", quality=q, subsampling=sub)
            ref = np.asarray(Image.open(stem + ".jpg").convert("RGB"))[..., ::-1].astype(np.float32)
            r, c = C.c_int(), C.c_int()
            out = np.zeros((h, w, 3), np.float32)
            assert host.apdhost_read_color_image(stem.encode(), C.byref(r), C.byref(c), out.ctypes.data_as(fp), out.size) == 0
            assert (r.value, c.value) == (h, w)
            assert np.array_equal(out, ref), (sub, q, int((out != ref).sum()))
    # a grey JPEG read as colour: three equal channels
    stem = str(tmp_path / "g")
    Image.fromarray(img[..., 0], "L").save(stem + ".jpg", quality=90)
    out = np.zeros((h, w, 3), np.float32)
    r, c = C.c_int(), C.c_int()
    assert host.apdhost_read_color_image(stem.encode(), C.byref(r), C.byref(c), out.ctypes.data_as(fp), out.size) == 0
    ref = np.asarray(Image.open(stem + ".jpg")).astype(np.float32)
    assert np.array_equal(out[..., 0], ref) and np.array_equal(out[..., 1], ref) and np.array_equal(out[..., 2], ref)


def test_malformed_jpegs_never_leave_their_buffers(tmp_path):
    """The decoder replaces libjpeg for every input image of the drop-in binary, so dataset files reach it directly: corrupt
    files may be refused or decoded to garbage, never read or written out of bounds.  The decoder is built with
    AddressSanitizer + UBSan (CPU) around tests/helpers/jpeg_fuzz.cpp and fed truncated files, random byte flips and the
    crafted SOS header whose Huffman table selectors (0..15 in the file) used to index dc[4] / ac[4] unchecked."""
    import subprocess
    Image = pytest.importorskip("PIL.Image")
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "jpeg_fuzz_asan")
    srcs = [os.path.join(here, "helpers", "jpeg_fuzz.cpp"), os.path.join(ROOT, "apd-mvs_amd", "host", "jpeg_gray.cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-fno-omit-frame-pointer"] + srcs + ["-o", exe])
    rng = np.random.RandomState(5)
    seeds = []
    for k, (mode, sub, size, extra) in enumerate([("L", 0, (40, 30), {}), ("RGB", 2, (37, 29), {}), ("RGB", 1, (48, 16), {}),
                                                   ("RGB", 0, (24, 24), {"optimize": True}), ("RGB", 2, (40, 24), {"progressive": True}),
                                                   ("L", 0, (33, 21), {"progressive": True})]):
        w, h = size
        img = (rng.rand(h, w, 3) * 255).astype(np.uint8)
        p = tmp_path / ("seed%d.jpg" % k)
        kw = dict(extra)
        if mode == "RGB":
            kw["subsampling"] = sub
        Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(p, quality=85, **kw)
        seeds.append(p.read_bytes())
    files = []
    must_refuse = []

    def emit(data):
        p = tmp_path / ("m%04d.jpg" % len(files))
        p.write_bytes(bytes(data))
        files.append(str(p))

    for data in seeds:
        emit(data)  # the intact file must decode
        sos = data.find(b"\xff\xda")
        assert sos > 0
        ns = data[sos + 4]
        for s in range(ns):  # table selectors beyond 3, one component at a time and all at once
            for val in (0x40, 0x04, 0xF0, 0x0F, 0xFF):
                d = bytearray(data)
                d[sos + 6 + 2 * s] = val
                emit(d)
        d = bytearray(data)
        for s in range(ns):
            d[sos + 6 + 2 * s] = 0xFF
        emit(d)
        for cut in (2, 10, sos, sos + 5, sos + 12, len(data) // 2, len(data) - 2):  # truncations
            emit(data[:cut])
        # a second frame header (it used to replace width / height / sampling under coefficient arrays and planes
        # sized for the first one -> heap overflow in the next scan, out-of-bounds read in the final copy): larger frame,
        # placed before the second scan, before the last scan and right before EOI
        sof = max(data.find(b"\xff\xc0"), data.find(b"\xff\xc2"))
        assert 0 < sof < sos
        seg = bytearray(data[sof:sof + 2 + int.from_bytes(data[sof + 2:sof + 4], "big")])
        seg[5:7] = (512).to_bytes(2, "big")
        seg[7:9] = (512).to_bytes(2, "big")
        scans = [i for i in range(sos, len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
        eoi = data.rfind(b"\xff\xd9")
        progressive = data[sof + 1] == 0xC2
        for at in sorted({scans[min(1, len(scans) - 1)], scans[-1], eoi}):
            if at <= scans[-1] or progressive:  # a sequential file is complete after its last scan: trailing markers are not read
                must_refuse.append(len(files))
            emit(data[:at] + bytes(seg) + data[at:])
        if len(scans) > 1:  # progressive: a file that stops after its first scans (EOI appended) is incomplete -> refused, not
            must_refuse.append(len(files))  # decoded without libjpeg's block smoothing
            emit(data[:scans[1]] + b"\xff\xd9")
        for _ in range(60):  # random flips in the headers (before the scan data) and in the entropy-coded segment
            d = bytearray(data)
            for _ in range(rng.randint(1, 4)):
                lo, hi = (2, sos + 14) if rng.rand() < 0.7 else (sos + 14, len(d))
                d[rng.randint(lo, hi)] = rng.randint(0, 256)
            emit(d)
    r = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, "sanitizer report:\n" + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    intact = [ln for ln in lines if "seed" not in ln and ln.split()[0].endswith(("m0000.jpg",))]
    assert intact and "grey=1 colour=1" in intact[0]
    refused = sum("grey=0" in ln for ln in lines)
    assert refused > 20, "the crafted selector / truncated files must be refused"
    assert len(must_refuse) >= 10
    for i in must_refuse:
        assert "grey=0 colour=0" in lines[i], "duplicate SOF / incomplete progressive file was decoded: " + lines[i]


@pytest.mark.parametrize("rows,cols,scale", [(4130, 6200, 8), (4130, 6200, 4), (4130, 6200, 2), (1080, 1920, 2), (517, 775, 2), (33, 47, 4)])
def test_resize_linear_matches_independent_bilinear_at_the_pyramid_ratios(host, rows, cols, scale):
    """ResizeLinear (host/APD.cpp) restates cv::resize(float, INTER_LINEAR) (APD.cpp:464-474: new size = round(old / scale), so
    the ratios are NOT integers: 4130 / 8 -> 516 rows).  OpenCV is not in the image; two independent implementations of the
    same definition -- half-pixel centres, taps floor(fx) and floor(fx) + 1 clamped to the image, no antialiasing -- are:
    scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True) (double arithmetic) and torch's bilinear interpolate
    (align_corners=False, antialias=False).  Agreement to float rounding at the sizes the four-level pyramid of a
    6200 x 4130 image really uses; the power-of-two special case is covered above."""
    import math
    import scipy.ndimage as ndi
    import torch
    rng = np.random.RandomState(rows + cols + scale)
    coarse = rng.rand(rows // 16 + 2, cols // 16 + 2) * 255
    a = np.kron(coarse, np.ones((16, 16)))[:rows, :cols].astype(np.float32)
    a += (rng.rand(rows, cols) * 8).astype(np.float32)
    new_rows = int(math.floor(rows / scale + 0.5))   # std::round of a positive value
    new_cols = int(math.floor(cols / scale + 0.5))
    out = np.zeros((new_rows, new_cols), np.float32)
    fp = C.POINTER(C.c_float)
    host.apdhost_resize_linear(a.ctypes.data_as(fp), rows, cols, out.ctypes.data_as(fp), new_rows, new_cols)
    ref1 = ndi.zoom(a.astype(np.float64), (new_rows / rows, new_cols / cols), order=1, mode="nearest", grid_mode=True)
    assert ref1.shape == out.shape
    ref2 = torch.nn.functional.interpolate(torch.from_numpy(a)[None, None].double(), size=(new_rows, new_cols), mode="bilinear",
                                           align_corners=False, antialias=False)[0, 0].numpy()
    assert np.abs(ref1 - ref2).max() < 1e-9          # the two references agree with each other
    # cv::resize keeps the source coordinate in binary32: fx = (float)((dx + 0.5) * scale - 0.5).  At x ~ 6000 one ulp of fx is
    # 4.9e-4 px, so against double coordinates the result may move by up to ~ulp/2 x local contrast (0.03 grey levels on this
    # deliberately blocky image) -- a property of the definition, not of this implementation ...
    assert np.abs(out - ref1).max() < 0.1 and np.abs(out - ref1).mean() < 5e-4
    # ... so the sharp check interpolates, in double and with another engine (scipy.ndimage.map_coordinates), at exactly the
    # binary32 coordinates OpenCV defines: what is left is the binary32 rounding of the three lerps
    fy = ((np.arange(new_rows) + 0.5) * (rows / new_rows) - 0.5).astype(np.float32).astype(np.float64)
    fx = ((np.arange(new_cols) + 0.5) * (cols / new_cols) - 0.5).astype(np.float32).astype(np.float64)
    yy, xx = np.meshgrid(np.clip(fy, 0, rows - 1), np.clip(fx, 0, cols - 1), indexing="ij")
    ref3 = ndi.map_coordinates(a.astype(np.float64), [yy, xx], order=1, mode="nearest")
    assert np.abs(out - ref3).max() < 1e-4, np.abs(out - ref3).max()
