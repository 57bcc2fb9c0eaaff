"""CPU analysis (oracle state): spread of the reliable neighbours of slot k over the 64 list entries of a K9/K10 wave (DESIGN.md
section 6: why LDS windows cannot serve the sub-patches).  Usage: python tools/nb_spread.py"""
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
t=time.time()
o = common.make_oracle(ob, sc, imgs, N, p1); o.run()
print("pass0", time.time()-t)
prior = common.postprocess(o.planes, o.weak_info, o.selected_views, p1["depth_min"], p1["depth_max"])
print("weak frac", (prior[2]==ob.WEAK).mean())
p2 = common.base_params(sc, N, seed=11, weak_peak_radius=6, state=ob.REFINE_INIT, use_APD=1, rotate_time=4, ransac_threshold=0.01-0.00125*3)
o2 = common.make_oracle(ob, sc, imgs, N, p2, prior=prior)
for k in (1,2,3,4): o2.run_kernel(k)
wi = o2.weak_info.copy(); nmap = o2.neighbours_map.copy(); nb = o2.neighbours.copy()
print("weak after K4", (wi==ob.WEAK).mean(), nb.shape)
# emulate compaction: colour 0 (black: (x+y)%2==0 ?), tiles 16x8
stats = {k: [] for k in range(8)}
allspan=[]
for colour in (0,1):
    lst=[]
    for ty in range(0,H,8):
        for tx in range(0,W,16):
            sub = wi[ty:ty+8, tx:tx+16]
            ys,xs = np.nonzero(sub==ob.WEAK)
            for y,x in zip(ys,xs):
                if ((x+tx)+(y+ty))%2==colour:
                    lst.append((x+tx,y+ty))
    lst=np.array(lst)
    for w0 in range(0,len(lst)-63,64):
        px = lst[w0:w0+64]
        idx = nmap[px[:,1],px[:,0]]
        q = nb[idx]  # [64, 9, 2]
        cspan = (px[:,0].max()-px[:,0].min(), px[:,1].max()-px[:,1].min())
        allspan.append(cspan)
        for k in range(8):
            qq = q[:,k+1]
            v = qq[:,0]>=0
            if v.sum()<2: continue
            stats[k].append((qq[v,0].max()-qq[v,0].min(), qq[v,1].max()-qq[v,1].min(), v.sum()))
allspan=np.array(allspan)
print("waves", len(allspan), "centre span x median/90%%: %d %d  y: %d %d" % (np.median(allspan[:,0]), np.percentile(allspan[:,0],90), np.median(allspan[:,1]), np.percentile(allspan[:,1],90)))
for k in range(8):
    s=np.array(stats[k])
    fit = ((s[:,0] <= 64-12) & (s[:,1] <= 32-12)).mean()
    fit21 = ((s[:,0] <= 64-12) & (s[:,1] <= 20-12)).mean()
    print("k=%d n=%d span x med %d p90 %d | y med %d p90 %d | fits 64x32: %.2f 64x20: %.2f" % (k, len(s), np.median(s[:,0]), np.percentile(s[:,0],90), np.median(s[:,1]), np.percentile(s[:,1],90), fit, fit21))
