"""Randomised check of the device fusion against the reference's sequential loop (oracle/fusion_oracle.cpp): random
sizes, view counts, source lists, noise levels, holes, colour or grey images, block masks; the PLY files must be
byte-identical.  Usage: python tools/fusion_fuzz.py [cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import pipeline, synth
from oracle import binding as ob
import test_gpu_dropin_binary as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = "/tmp/fusion_fuzz"
os.makedirs(out, exist_ok=True)
bad = 0
t0 = time.time()
for case in range(first, first + cases):
    rng = np.random.RandomState(5000 + case)
    W, H = int(rng.randint(24, 400)), int(rng.randint(20, 300))
    V = int(rng.randint(2, 9))
    S = int(rng.randint(1, V))
    noise = float(rng.choice([0.0, 0.0003, 0.001, 0.004]))
    scene, results = T._fusion_inputs(synth, pipeline, pkg, W, H, V, S, noise, seed=case)
    for v in range(V):  # random source order, extra holes
        rng.shuffle(scene.pairs[v])
        results[v].depth[rng.rand(H, W) < rng.choice([0.0, 0.1, 0.4])] = 0.0
    colour = None
    if rng.rand() < 0.5:
        colour = [np.ascontiguousarray(np.stack([im, 255.0 - im, np.roll(im, 5, 0)], -1), np.float32) for im in scene.images]
    masks = None
    if rng.rand() < 0.4:
        masks = [(rng.rand(H, W) < 0.7).astype(np.uint8) * 255 for _ in range(V)]
    cams = (type(scene.cameras[0]) * V)(*scene.cameras)
    n_cpu = ob.fuse(cams, colour if colour is not None else scene.images, [results[v].depth for v in range(V)],
                    [results[v].normal for v in range(V)], [results[v].weak for v in range(V)], scene.pairs,
                    os.path.join(out, "cpu.ply"), blocks=masks)
    n_gpu = pipeline.fuse(scene, results, os.path.join(out, "gpu.ply"), colour_images=colour, block_masks=masks)
    same = open(os.path.join(out, "cpu.ply"), "rb").read() == open(os.path.join(out, "gpu.ply"), "rb").read()
    label = "case %d: %dx%d views=%d sources=%d noise=%.4f %s%s points=%d" % (case, W, H, V, S, noise, "colour" if colour else "grey",
                                                                             " masks" if masks else "", n_gpu)
    if same and n_cpu == n_gpu:
        print("ok   " + label, flush=True)
    else:
        bad += 1
        print("FAIL " + label + " (host loop: %d points)" % n_cpu, flush=True)
print("%d case(s), %d failure(s), %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
