import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from glio_amd import ctypes_types as T
from oracle import pyoracle as po
win = synth.make_window(W=3, pts_per_scan=4000, seed=synth.SEED_BASE + 5)
ctx = capi.Context(win.opts); ctx.set_map(win.map_pts)
s = 2; i = 3545
q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
ctx.associate(s, win.scans[s], q2, t2)
hnn = np.zeros((len(win.scans[s]), 5), np.int32)
capi.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
f = hnn[i].view(np.float32)
print("gpu p", [repr(x) for x in f[:3]], "d4th", repr(f[3]), "d5th", repr(f[4]))
print("gpu p bits", [hex(x & 0xffffffff) for x in hnn[i][:3]])
