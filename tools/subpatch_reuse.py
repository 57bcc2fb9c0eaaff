"""CPU analysis (oracle state, float64 geometry): how often the 72 sub-patch samples of a WEAK pixel fall on the same texel quad
under consecutive / earlier hypotheses of the propagation phase.  Usage: python tools/subpatch_reuse.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"oracle"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from apd_mvs_amd import synth
import binding as ob
import common
W,H,N = 768,576,4
sc, imgs = common.scene_inputs(synth, W, H, N, seed=3, textureless=0.2)
p1 = common.base_params(sc, N, seed=11, weak_peak_radius=6, max_iterations=3)
o = common.make_oracle(ob, sc, imgs, N, p1); o.run()
prior = common.postprocess(o.planes, o.weak_info, o.selected_views, p1["depth_min"], p1["depth_max"])
p2 = common.base_params(sc, N, seed=11, weak_peak_radius=6, state=ob.REFINE_INIT, use_APD=1, rotate_time=4, ransac_threshold=0.01-0.00125*3, max_iterations=3)
o2 = common.make_oracle(ob, sc, imgs, N, p2, prior=prior)
for k in (1,2,3,4,5): o2.run_kernel(k)
def stats(tag):
    wi = o2.weak_info; nmap = o2.neighbours_map; nb = o2.neighbours; planes = o2.planes.astype(np.float64)  # camera-frame planes (n, w) during the sweep
    K = [np.array(sc.K[i], np.float64).reshape(3,3) for i in range(N+1)]
    R = [np.array(sc.R[i], np.float64).reshape(3,3) for i in range(N+1)]
    t = [np.array(sc.t[i], np.float64).reshape(3) for i in range(N+1)]
    C = [-R[i].T @ t[i] for i in range(N+1)]
    ys, xs = np.nonzero(wi == ob.WEAK)
    rng = np.random.RandomState(0)
    sel = rng.choice(len(ys), min(3000, len(ys)), replace=False)
    same_prev = 0; same_any = 0; total = 0
    for idx in sel:
        y, x = ys[idx], xs[idx]
        q = nb[nmap[y, x]][1:]
        valid = [(int(a), int(b)) for a, b in q if a >= 0 and wi[b, a] == ob.STRONG]
        hyps = [planes[b, a] for a, b in valid] + [planes[y, x]]
        nbs = [(int(a), int(b)) for a, b in q if a >= 0]
        for v in range(1, N+1):
            Rr = R[v] @ R[0].T; tr = R[v] @ (C[0] - C[v])
            addr = []
            for pl in hyps:
                qv = pl[:3] / pl[3]
                M = Rr - np.outer(tr, qv)
                Hm = K[v] @ M @ np.linalg.inv(K[0])
                a_h = []
                for (nx_, ny_) in nbs:
                    pts = np.array([[nx_ + 5*i, ny_ + 5*j, 1.0] for i in (-1,0,1) for j in (-1,0,1)]).T
                    pr = Hm @ pts
                    X = np.floor(pr[0]/pr[2]).astype(int); Y = np.floor(pr[1]/pr[2]).astype(int)
                    a_h.append(Y * 100000 + X)
                addr.append(np.concatenate(a_h) if a_h else np.zeros(0, int))
            addr = np.array(addr)   # [hyps, samples]
            if addr.size == 0: continue
            for h in range(1, addr.shape[0]):
                same_prev += int((addr[h] == addr[h-1]).sum())
                same_any += int(np.any(addr[:h] == addr[h][None, :], axis=0).sum())
                total += addr.shape[1]
    print(tag, "weak px sampled", len(sel), "same as previous hyp: %.3f  same as any earlier: %.3f" % (same_prev/total, same_any/total))
stats("after K5 (iteration 0)")
o2.run_sweeps(0, 1)
stats("after iteration 0")
o2.run_sweeps(1, 1)
stats("after iteration 1")
