import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from glio_amd import ctypes_types as T
from oracle import pyoracle as po
win = synth.make_window(W=3, pts_per_scan=4000, seed=synth.SEED_BASE + 5)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
M = win.map_pts
for s in range(win.W):
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
    pts, pl, sc, src, nn = po.associate(win.opts, M, win.scans[s], q2, t2, want_nn=True)
    cnt = ctx.associate(s, win.scans[s], q2, t2)
    hnn = np.zeros((len(win.scans[s]), 5), np.int32)
    capi.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
    bad = np.where((hnn != nn).any(axis=1) & (hnn[:, 4] >= 0))[0]
    print("slot", s, "mismatch rows", len(bad))
    R = synth.q2R(q2 / np.linalg.norm(q2))
    for i in bad[:5]:
        p = (R @ win.scans[s][i, :3].astype(float) + t2).astype(np.float32)
        def d(idx): 
            e = p - M[idx, :3]; 
            return np.float32(np.float32(e[0]*e[0]) + np.float32(e[1]*e[1])) + np.float32(e[2]*e[2])
        print("  row", i, "p", p, "\n   oracle", nn[i], [float(d(k)) for k in nn[i]], "\n   hip   ", hnn[i], [float(d(k)) for k in hnn[i] if k >= 0])
        print("   cells", [tuple(np.floor(M[k,:3]*np.float32(0.8)).astype(int)) for k in nn[i]], "q cell", tuple(np.floor(p*np.float32(0.8)).astype(int)))
