"""numpy statistics behind the round-5 K2 design (no GPU): for the C2 association workload, how many (query, candidate) pairs the 27-cell
search ranks, how many the 4x4x4 half-cell NEAR block would, and how many queries the near block certifies (fifth neighbour found inside it closer
than the query's distance to the block boundary => the five are the global five)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from scipy.spatial import cKDTree
from glio_amd import synth
from glio_amd.capi import lidar_pose

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
cell = np.float32(1.25); half = cell / 2
mp = win.map_pts[:, :3].astype(np.float64)
tree = cKDTree(mp)
mh = np.floor(mp / half).astype(np.int64)              # half-cell index of the map points
print("map points", len(mp))
for s in range(0, 20, max(1, 20 // W)):
    q2, t2 = lidar_pose(win.opts, win.init.quat[s], win.init.trans[s])
    R = synth.q2R(q2)
    p = (win.scans[s][:, :3].astype(np.float64) @ R.T + t2).astype(np.float32).astype(np.float64)
    d, idx = tree.query(p, k=5)
    d5 = d[:, 4]
    c = np.floor(p / cell)
    f = p - c * cell
    b = half + np.minimum(f, cell - f).min(axis=1) - 1e-3
    ok = d5 < b
    gate = d5 * d5 < 1.5
    # units: queries by cell
    key = (c[:, 0].astype(np.int64) << 42) + (c[:, 1].astype(np.int64) << 21) + c[:, 2].astype(np.int64)
    uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    units16 = np.ceil(cnt / 16).sum()
    fail_cells = np.unique(inv[~ok]).size
    # candidates of the near block per cell: half-cells [2c-1, 2c+2]^3
    cu = np.array([c[np.argmax(inv == k)] for k in range(0, len(uk), max(1, len(uk) // 400))])
    nb, full = [], []
    for cc in cu:
        lo = 2 * cc - 1
        inb = np.all((mh >= lo) & (mh <= lo + 3), axis=1)
        nb.append(inb.sum())
        mc = np.floor(mp / cell)
        full.append(np.all(np.abs(mc - cc) <= 1, axis=1).sum())
    nb, full = np.array(nb), np.array(full)
    print(f"slot {s}: cells {len(uk)} units16 {int(units16)} q/cell {cnt.mean():.1f}; near-block cand mean {nb.mean():.1f} p50 {np.median(nb):.0f} p90 {np.percentile(nb, 90):.0f} "
          f"p99 {np.percentile(nb, 99):.0f} max {nb.max()}; 27-cell cand mean {full.mean():.1f}; certified {ok.mean():.4f} (of gate-passing {ok[gate].mean():.4f}), gate pass {gate.mean():.4f}; "
          f"cells with a failure {fail_cells / len(uk):.3f}; d5 median {np.median(d5):.3f} p90 {np.percentile(d5, 90):.3f}")
