#!/usr/bin/env python3
"""What changes when the views of a geometric pass are processed as a device list processes them (Jacobi: every view reads the depth
maps its sources had after the PREVIOUS pass; host/multi_device.cpp, `APD folder 0 --jacobi` = what `APD folder 0,1,...,7` computes)
instead of in the reference's order (Gauss-Seidel: a view reads what its earlier sources wrote in THIS pass; main.cpp:117-124,
APD.cpp:497-509).  SURVEY 8(e) prescribes statistics, not bits: per view the share of pixels within 1e-3 relative depth / 1 degree of
normal of the reference-order result, the agreement of the WEAK / STRONG / UNKNOWN maps, BOTH orders against the analytic ground
truth (depth and normal), and the fused point counts.  A third run is the control: the reference's order again with another RNG seed
(the reference seeds with clock64(), APD.cu:803: two runs of the reference itself differ like that) -- what the order of views changes
has to be read against what the algorithm's own randomness changes.  Normals are compared on textured pixels (in the textureless
rectangles of the scene the photometric cost does not constrain them).

Usage: python tools/jacobi_vs_gs.py W H views src [--hard] [--json out.json] [extra APD flags]
Prints one line per view and a summary; --json writes the summary (tests/test_gpu_dropin_binary.py asserts floors on it)."""
import json
import os
import shutil
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APD = os.path.join(ROOT, "apd-mvs_amd", "_build", "APD")


def read_dmb(path):
    raw = open(path, "rb").read()
    version, rows, cols, typ = struct.unpack("<4i", raw[:16])
    dt, ch = {5: (np.float32, 1), 21: (np.float32, 3), 0: (np.uint8, 1), 4: (np.uint32, 1)}[typ]
    a = np.frombuffer(raw[16:], dt)
    return a.reshape(rows, cols, ch) if ch > 1 else a.reshape(rows, cols)


def run(W, H, V, S, hard=False, extra=(), work="/tmp/jvg", seed=12345, quiet=False):
    base = work + "_base"
    shutil.rmtree(base, ignore_errors=True)
    os.makedirs(base)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_dense.py"), base, "--width", str(W), "--height", str(H),
                           "--views", str(V), "--src", str(S), "--textureless", "0.2", "--jpeg", "--gt"] + (["--hard"] if hard else []),
                          stdout=subprocess.DEVNULL)
    out, wall = {}, {}
    for name, flags, sd in (("gs", [], seed), ("jacobi", ["--jacobi"], seed), ("gs_other_seed", [], seed + 1)):
        d = "%s_%s" % (work, name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(base, d)
        r = subprocess.run([APD, d, "0", "--seed", str(sd), "--keep-maps"] + flags + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("APD %s failed:\n%s" % (name, r.stdout[-3000:]))
        wall[name] = [ln for ln in r.stdout.splitlines() if "passes" in ln.lower() and "ms" in ln][-1:] or [""]
        out[name] = d
    rows = []

    def angle(a, b):
        return np.degrees(np.arccos(np.clip((a * b).sum(-1), -1.0, 1.0)))

    for i in range(V):
        gt = np.load(os.path.join(base, "gt", "%08d.npy" % i))
        gt_n = np.load(os.path.join(base, "gt", "%08d_normal.npy" % i))
        textured = np.load(os.path.join(base, "gt", "%08d_textured.npy" % i))
        m = {}
        for name in out:
            f = os.path.join(out[name], "APD", "%08d" % i)
            m[name] = (read_dmb(os.path.join(f, "depths.dmb")), read_dmb(os.path.join(f, "normals.dmb")), read_dmb(os.path.join(f, "weak.bin")))
        (da, na, wa) = m["gs"]
        row = {"view": i, "valid_gs": float((da > 0).mean()), "states_gs": np.bincount(wa.ravel(), minlength=3).tolist()}
        for other in ("jacobi", "gs_other_seed"):   # the order of views, and the control: the same order with another seed
            (db, nb, wb) = m[other]
            both = (da > 0) & (db > 0)
            rel = np.abs(da[both] - db[both]) / da[both]
            ang = angle(na[both], nb[both])
            tex = textured[both]
            row.update({"%s:both_valid" % other: float(both.mean()),
                        "%s:depth_within_1e-3" % other: float((rel <= 1e-3).mean()), "%s:depth_within_1e-2" % other: float((rel <= 1e-2).mean()),
                        "%s:normal_within_1deg_textured" % other: float((ang[tex] <= 1.0).mean()),
                        "%s:normal_within_5deg_textured" % other: float((ang[tex] <= 5.0).mean()),
                        "%s:normal_median_deg_textured" % other: float(np.median(ang[tex])),
                        "%s:weak_map_agreement" % other: float((wa == wb).mean())})
        inner = np.zeros_like(gt, bool)
        inner[8:-8, 8:-8] = True
        for name in out:
            d, n = m[name][0], m[name][1]
            err = np.abs(d - gt) / gt
            row["gt:depth_within_1e-2:%s" % name] = float((err[inner] <= 1e-2).mean())    # of all interior pixels (an invalid pixel is a miss)
            row["gt:depth_within_1e-3:%s" % name] = float((err[inner] <= 1e-3).mean())
            okn = inner & textured & (d > 0)
            a = angle(n[okn], gt_n[okn])
            row["gt:normal_within_5deg_textured:%s" % name] = float((a <= 5.0).mean())
            row["gt:normal_median_deg_textured:%s" % name] = float(np.median(a))
        rows.append(row)
        if not quiet:
            print("view %2d: jacobi vs gs: depth 1e-3 %.4f, normal 1deg %.4f (median %.2f deg), weak maps equal %.4f | control (gs, other seed) vs gs: depth 1e-3 %.4f, "
                  "normal 1deg %.4f (median %.2f deg), weak maps equal %.4f | ground truth, depth 1e-2: gs %.4f jacobi %.4f; normal median deg: gs %.2f jacobi %.2f" % (
                      i, row["jacobi:depth_within_1e-3"], row["jacobi:normal_within_1deg_textured"], row["jacobi:normal_median_deg_textured"],
                      row["jacobi:weak_map_agreement"], row["gs_other_seed:depth_within_1e-3"], row["gs_other_seed:normal_within_1deg_textured"],
                      row["gs_other_seed:normal_median_deg_textured"], row["gs_other_seed:weak_map_agreement"], row["gt:depth_within_1e-2:gs"],
                      row["gt:depth_within_1e-2:jacobi"], row["gt:normal_median_deg_textured:gs"], row["gt:normal_median_deg_textured:jacobi"]))
    points = {}
    for name, d in out.items():
        head = open(os.path.join(d, "APD", "APD.ply"), "rb").read(400).split(b"end_header\n", 1)[0].decode()
        points[name] = int([ln for ln in head.split("\n") if ln.startswith("element vertex")][0].split()[-1])
    keys = [k for k in rows[0] if isinstance(rows[0][k], float)]
    summary = {"config": {"width": W, "height": H, "views": V, "src": S, "hard": bool(hard), "seed": seed, "extra": list(extra)},
               "mean": {k: float(np.mean([r[k] for r in rows])) for k in keys}, "min": {k: float(np.min([r[k] for r in rows])) for k in keys},
               "fused_points": points, "per_view": rows}
    if not quiet:
        print("%-48s %10s %10s" % ("over %d views" % V, "mean", "min"))
        for k in keys:
            print("%-48s %10.4f %10.4f" % (k, summary["mean"][k], summary["min"][k]))
        print("fused points: gs %d, jacobi %d (%+.3f %%), gs with another seed %d (%+.3f %%)" % (
            points["gs"], points["jacobi"], 100.0 * (points["jacobi"] - points["gs"]) / max(points["gs"], 1), points["gs_other_seed"],
            100.0 * (points["gs_other_seed"] - points["gs"]) / max(points["gs"], 1)))
    for d in list(out.values()) + [base]:
        shutil.rmtree(d, ignore_errors=True)
    return summary


if __name__ == "__main__":
    args = sys.argv[1:]
    hard = "--hard" in args
    if hard:
        args.remove("--hard")
    jpath = None
    if "--json" in args:
        k = args.index("--json")
        jpath = args[k + 1]
        del args[k:k + 2]
    W, H, V, S = (int(v) for v in args[:4])
    s = run(W, H, V, S, hard=hard, extra=args[4:])
    if jpath:
        with open(jpath, "w") as f:
            json.dump(s, f)
