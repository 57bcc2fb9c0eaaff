import os, sys, subprocess, numpy as np, shutil
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tools")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
import make_synthetic_dense as msd
from PIL import Image
import test_gpu_dropin_binary as T
a, b = "/tmp/ce_a", "/tmp/ce_b"
for d in (a, b):
    shutil.rmtree(d, ignore_errors=True)
msd.write_dense_folder(a, synth, 320, 240, 5, 4, seed=1, jpeg=False)
# colourise: replace the pgm files by colour JPEGs whose luma is (close to) the grey image
for i in range(5):
    g = np.frombuffer(open(os.path.join(a, "images", "%08d.pgm" % i), "rb").read()[-320*240:], np.uint8).reshape(240, 320).astype(np.float32)
    rgb = np.stack([np.clip(g * 1.1, 0, 255), g, np.clip(g * 0.8 + 20, 0, 255)], -1).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(os.path.join(a, "images", "%08d.jpg" % i), quality=95, subsampling=2)
    os.remove(os.path.join(a, "images", "%08d.pgm" % i))
shutil.copytree(a, b)
r = subprocess.run([os.path.join(ROOT, "apd-mvs_amd/_build/APD"), a, "0", "--seed", "3", "--iters", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-200:])
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/mvs_pipeline.py"), b, "--seed", "3", "--iters", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-200:])
pa, pb = open(os.path.join(a, "APD/APD.ply"), "rb").read(), open(os.path.join(b, "APD/APD.ply"), "rb").read()
xyz, bgr = T._read_ply(os.path.join(a, "APD/APD.ply"))
print("identical:", pa == pb, "points:", len(xyz), "mean BGR:", bgr.mean(0))
