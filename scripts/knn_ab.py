"""A/B timing of the K2 neighbour search: tiled (mode 0, default) vs one 16-lane group per query (mode 1), on the bench workloads:
one 64k scan of C2, the one-call window association (20 x 64k), C3 (131k queries vs a 1.19M-point map), the batch pair association."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glio_amd import batch, capi, synth          # noqa: E402
from glio_amd.capi import lidar_pose             # noqa: E402

lib = capi.load()
out = {}
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
big = synth.tiled_map(synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7).map_pts, 24)
w3 = synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7)
K, pts, sr = 16, 32768, 6
wb = synth.make_window(W=K, pts_per_scan=pts, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=25.0, map_density=0.5)
for mode in (2, 3, 0, 2, 3, 0):
    lib.glio_debug_set_knn_mode(mode)
    r = {}
    ctx = capi.Context(win.opts)
    ctx.set_map(win.map_pts)
    for s in range(win.W):
        ctx.set_scan(s, win.scans[s])
    poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
    cnt0 = ctx.associate_resident(0, *poses[0])
    r["scan_us"] = round(min(ctx.time_kernel(capi.KERNEL_ASSOCIATE, 20) for _ in range(3)) * 1e3, 1)
    q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
    ctx.associate_window(q2s, t2s)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); cn = ctx.associate_window(q2s, t2s); ts.append(time.perf_counter() - t0)
    r["window_ms"] = round(min(ts) * 1e3, 3); r["kept"] = int(np.sum(cn)); r["kept0"] = int(cnt0)
    ctx.close()
    o = synth.default_opts(1, pts=131072, map_pts=len(big))
    c3 = capi.Context(o)
    c3.set_map(big)
    q2, t2 = lidar_pose(o, w3.init.quat[0], w3.init.trans[0])
    r["c3_kept"] = int(c3.associate(0, w3.scans[0], q2, t2))
    r["c3_us"] = round(min(c3.time_kernel(capi.KERNEL_ASSOCIATE, 10) for _ in range(3)) * 1e3, 1)
    c3.close()
    tlb = np.array(wb.opts.t_lb, np.float32)
    bposes = np.c_[wb.init.trans, wb.init.quat]
    ci, cj = batch.pair_list(K, sr)
    ba = batch.BatchAssociation(K, pts, int(len(ci)) * pts)
    for k in range(K):
        sc = wb.scans[k].copy(); sc[:, :3] -= tlb
        ba.set_frame(k, sc)
    ba.run(bposes, ci, cj)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); counts, total = ba.run(bposes, ci, cj); ts.append(time.perf_counter() - t0)
    r["pairs_ms"] = round(min(ts) * 1e3, 2); r["pairs_kept"] = int(total)
    ba.close()
    out.setdefault(f"mode{mode}", []).append(r)
    print(mode, r, flush=True)
json.dump(out, open("gpurun_out/knn_ab.json", "w"), indent=1)
