import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from oracle import pyoracle as po
win = synth.make_window(W=4, pts_per_scan=600, with_gnss=True, with_prior=True, seed=synth.SEED_BASE)
corr = []
for s in range(win.W):
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
    pts, pl, sc, _ = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
    corr.append((pts, pl, sc))
corr[1] = (corr[1][0][:0], corr[1][1][:0], corr[1][2][:0])
corr[2] = (corr[2][0][:1], corr[2][1][:1], corr[2][2][:1])
corr[3] = (corr[3][0][:257], corr[3][1][:257], corr[3][2][:257])
prob = po.Problem(win, corr)
def run(tag):
    ctx = capi.Context(win.opts); ctx.load_window(win, corr)
    Ho, go, co = prob.linearize(win.init); Hh, gh, ch = ctx.linearize(win.init)
    d = np.abs(Hh - Ho)
    blk = np.array([[d[15*a:15*a+15, 15*b:15*b+15].max() for b in range(win.W)] for a in range(win.W)])
    print(tag, "cost oracle", co, "hip", ch, "max block diff\n", blk, "ddt part", d[60:, :].max() if d.shape[0] > 60 else None)
    ctx.close()
run("before")
# something like the assoc tests
w2 = synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7)
c2 = capi.Context(w2.opts); c2.set_map(w2.map_pts)
q2, t2 = po.lidar_pose_for_association(w2.opts, w2.init.quat[0], w2.init.trans[0])
print("assoc", c2.associate(0, w2.scans[0], q2, t2)); c2.close()
run("after")
