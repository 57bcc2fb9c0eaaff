"""CPU analysis (oracle state): could lanes of one K9/K10 wave share sub-patch costs or lines?  Counts, per 16 x 8 tile of one colour, the
distinct (neighbour position, candidate plane) pairs among its WEAK pixels and the distinct neighbour positions per slot (K3 order, sorted by
position, sorted by direction).  DESIGN.md section 6.  Usage: python tools/pair_dup_stats.py"""
import sys, os
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
wi = o2.weak_info; nmap = o2.neighbours_map; nb = o2.neighbours
print("weak px", int((wi==ob.WEAK).sum()), "of", W*H)
tot_pairs = 0; distinct_pairs = 0; tot_q = 0; distinct_q = 0; groups = 0
for colour in (0,1):
    for ty in range(0, H, 8):
        for tx in range(0, W, 16):
            pairs = set(); qs = set(); n_pairs = 0; n_q = 0
            for y in range(ty, min(ty+8,H)):
                for x in range(tx, min(tx+16,W)):
                    if ((x+y)&1) != colour or wi[y,x] != ob.WEAK: continue
                    q = nb[nmap[y,x]][1:]
                    nbs = [(int(a),int(b)) for a,b in q if a >= 0]
                    hyps = [(int(a),int(b)) for a,b in q if a >= 0 and wi[b,a]==ob.STRONG]
                    for k in nbs:
                        qs.add(k); n_q += 1
                        for h in hyps:
                            pairs.add((k,h)); n_pairs += 1
            if n_pairs:
                groups += 1
                tot_pairs += n_pairs; distinct_pairs += len(pairs); tot_q += n_q; distinct_q += len(qs)
print("tiles with weak px", groups, "pairs (nbr k, candidate h) total", tot_pairs, "distinct", distinct_pairs, "ratio %.3f" % (distinct_pairs/tot_pairs))
print("neighbour positions total", tot_q, "distinct", distinct_q, "ratio %.3f" % (distinct_q/tot_q))

# distinct neighbour positions per slot per wave (16x8 tile of one colour): K3 order vs sorted by position
import itertools
def slot_stats(sort_mode):
    tot_slots = 0; tot_distinct = 0; tot_lanes = 0
    for colour in (0,1):
        for ty in range(0, H, 8):
            for tx in range(0, W, 16):
                per_slot = [set() for _ in range(8)]; cnt = [0]*8
                for y in range(ty, min(ty+8,H)):
                    for x in range(tx, min(tx+16,W)):
                        if ((x+y)&1) != colour or wi[y,x] != ob.WEAK: continue
                        q = [(int(a),int(b)) for a,b in nb[nmap[y,x]][1:]]
                        if sort_mode == 1:
                            q = sorted(q, key=lambda t: (t[1], t[0]))
                        elif sort_mode == 2:   # by direction angle from the pixel
                            import math
                            q = sorted(q, key=lambda t: (-9 if t[0] < 0 else math.atan2(t[1]-y, t[0]-x)))
                        for k,(a,b) in enumerate(q):
                            if a >= 0:
                                per_slot[k].add((a,b)); cnt[k] += 1
                for k in range(8):
                    if cnt[k]:
                        tot_slots += 1; tot_distinct += len(per_slot[k]); tot_lanes += cnt[k]
    print("sort_mode", sort_mode, "lanes per (tile,slot) %.1f distinct positions %.1f  ratio %.3f" % (tot_lanes/tot_slots, tot_distinct/tot_slots, tot_distinct/tot_lanes))
for m in (0,1,2): slot_stats(m)
