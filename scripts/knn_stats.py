"""Statistics of the near-block search (glio_debug_knn_stats) on the bench workloads: how many units / queries are certified, handed on, staged."""
import ctypes as C
import sys
sys.path.insert(0, ".")
import numpy as np
from glio_amd import capi, synth, batch
from glio_amd.capi import lidar_pose
lib = capi.load()
names = ["units", "over_units", "unit_entries", "uncertified_q", "queries", "certified_q", "staged", "scan_quads"]
def stats():
    o = (C.c_ulonglong * 8)()
    lib.glio_debug_knn_stats(1, o)
    return dict(zip(names, list(o)))
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
for s in range(win.W):
    ctx.set_scan(s, win.scans[s])
poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
stats()
for s in (0, 6, 12, 18):
    ctx.associate_resident(s, *poses[s])
    print("C2 slot", s, stats())
q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
ctx.associate_window(q2s, t2s)
print("C2 window", stats())
ctx.close()
big = synth.tiled_map(synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7).map_pts, 24)
w3 = synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7)
o = synth.default_opts(1, pts=131072, map_pts=len(big))
c3 = capi.Context(o); c3.set_map(big)
q2, t2 = lidar_pose(o, w3.init.quat[0], w3.init.trans[0])
c3.associate(0, w3.scans[0], q2, t2)
print("C3", stats())
c3.close()
K, pts, sr = 16, 32768, 6
wb = synth.make_window(W=K, pts_per_scan=pts, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=25.0, map_density=0.5)
tlb = np.array(wb.opts.t_lb, np.float32)
bposes = np.c_[wb.init.trans, wb.init.quat]
ci, cj = batch.pair_list(K, sr)
ba = batch.BatchAssociation(K, pts, int(len(ci)) * pts)
for k in range(K):
    sc = wb.scans[k].copy(); sc[:, :3] -= tlb
    ba.set_frame(k, sc)
stats()
ba.run(bposes, ci, cj)
print("pairs", len(ci), stats())
ba.close()
lib.glio_debug_knn_stats(0, None)
