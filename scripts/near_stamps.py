"""Per-phase device-clock stamps of k_knn5_near (library built with -DGLIO_DEV_STAMPS, GLIO_HIP_LIB=...): single C2 scan and the window call."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from glio_amd import synth, capi
from glio_amd.capi import lidar_pose
lib = capi.load()
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
ctx = capi.Context(win.opts); ctx.set_map(win.map_pts)
for s in range(win.W): ctx.set_scan(s, win.scans[s])
poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
def rd(tag, n):
    buf = np.zeros((n, 8), np.int64)
    lib.glio_debug_near_stamps(buf.ctypes.data_as(C.c_void_p), n)
    a = buf[buf[:, 5] > 0]
    ph = a[:, :5].sum(0) / len(a) / 100.0
    life = (a[:, 7] - a[:, 6]) / 100.0
    span = (buf[:, 7].max() - buf[buf[:, 6] > 0][:, 6].min()) / 100.0
    print(tag, "working waves %d of %d; mean us per wave: probe %.2f, runs+marks %.2f, staging %.2f, scan %.2f, rerank..hand-on %.2f; life mean %.2f p90 %.2f max %.2f; launch span %.1f us" % (
        len(a), n, ph[0], ph[1], ph[2], ph[3], ph[4], life.mean(), np.percentile(life, 90), life.max(), span))
ctx.associate_resident(0, *poses[0])
ctx.associate_resident(0, *poses[0]); rd("scan0", 2048)
ctx.associate_resident(12, *poses[12]); rd("scan12", 2048)
q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
ctx.associate_window(q2s, t2s)
ctx.associate_window(q2s, t2s); rd("window", 41024)
