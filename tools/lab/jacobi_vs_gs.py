#!/usr/bin/env python3
"""Jacobi (host/multi_device.cpp, `APD folder 0 --jacobi`) against Gauss-Seidel (the reference's order, `APD folder 0`) on one
synthetic dense folder: per-view valid fraction, WEAK/STRONG/UNKNOWN counts, closeness of the depth maps, fused points.
Usage: python tools/lab/jacobi_vs_gs.py W H views src [extra APD flags]"""
import os, shutil, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, H, V, S = (int(v) for v in sys.argv[1:5])
extra = sys.argv[5:]
APD = os.path.join(ROOT, "apd-mvs_amd", "_build", "APD")


def read_dmb(path):
    raw = open(path, "rb").read()
    version, rows, cols, typ = struct.unpack("<4i", raw[:16])
    dt, ch = {5: (np.float32, 1), 21: (np.float32, 3), 0: (np.uint8, 1), 4: (np.uint32, 1)}[typ]
    a = np.frombuffer(raw[16:], dt)
    return a.reshape(rows, cols, ch) if ch > 1 else a.reshape(rows, cols)


base = "/tmp/jvg_base"
shutil.rmtree(base, ignore_errors=True)
os.makedirs(base)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_dense.py"), base, "--width", str(W), "--height", str(H),
                       "--views", str(V), "--src", str(S), "--textureless", "0.2", "--jpeg"], stdout=subprocess.DEVNULL)
out = {}
for name, flags in (("gs", []), ("jacobi", ["--jacobi"])):
    d = "/tmp/jvg_" + name
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(base, d)
    r = subprocess.run([APD, d, "0", "--seed", "12345", "--keep-maps"] + flags + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(name, "rc", r.returncode, [l for l in r.stdout.splitlines() if "used" in l.lower() or "Round nums" in l][-2:])
    out[name] = d
for i in range(V):
    a = read_dmb(os.path.join(out["gs"], "APD", "%08d" % i, "depths.dmb"))
    b = read_dmb(os.path.join(out["jacobi"], "APD", "%08d" % i, "depths.dmb"))
    wa = read_dmb(os.path.join(out["gs"], "APD", "%08d" % i, "weak.bin"))
    wb = read_dmb(os.path.join(out["jacobi"], "APD", "%08d" % i, "weak.bin"))
    ok = (a > 0) & (b > 0)
    rel = np.abs(a[ok] - b[ok]) / b[ok]
    print("view %d: valid gs %.4f jacobi %.4f | states gs %s jacobi %s | both valid %.4f, within 1e-3 %.4f, 1e-2 %.4f, 2e-2 %.4f" % (
        i, (a > 0).mean(), (b > 0).mean(), np.bincount(wa.ravel(), minlength=3), np.bincount(wb.ravel(), minlength=3), ok.mean(),
        (rel <= 1e-3).mean(), (rel <= 1e-2).mean(), (rel <= 2e-2).mean()))
for name, d in out.items():
    raw = open(os.path.join(d, "APD", "APD.ply"), "rb").read()
    head = raw.split(b"end_header\n", 1)[0].decode()
    print(name, [l for l in head.split("\n") if l.startswith("element vertex")])
