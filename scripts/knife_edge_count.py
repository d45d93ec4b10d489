"""How often does the DEVICE-built state of a stream (local map voxel-averaged on the device with fixed-point sums, prior from the
device's Cholesky root) change an association record with respect to the ORACLE-built state (PCL-style float voxel grid, eigen-root
prior)?  On the same map the device association is bit-exact (tests/test_hip_assoc.py); here each side builds its own map and its
own prior chain over a stream of keyframes, and the kept correspondences are compared record by record.
    [GLIO_LOCALMAP_ACC=1] python scripts/knife_edge_count.py [keyframes] [points_per_scan] > profiles/r04_knife_edge[_float].json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from glio_amd import capi, sliding, synth  # noqa: E402
from glio_amd import ctypes_types as T  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_hip_streaming import OracleBackend  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 34
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
W = 5
long = synth.make_window(W=L, pts_per_scan=pts, seed=synth.SEED_BASE + 43)
opts = synth.default_opts(W, pts=max(1024, pts), map_pts=1 << 18)
tlb = np.array(opts.t_lb, np.float32)
first = T.WindowState(W)
first.trans[:], first.quat[:], first.speed_bias[:] = long.init.trans[:W], long.init.quat[:W], long.init.speed_bias[:W]
ctx = capi.Context(opts)
ctx.localmap_config(50, 0.4, pts)
ACC = int(os.environ.get("GLIO_LOCALMAP_ACC", "0"))     # 1: pcl::VoxelGrid's float sums in concatenation order (glio_localmap_set_accumulation)
ctx.localmap_set_accumulation(ACC)
po.lib().orc_set_assoc_grid.restype = None
po.lib().orc_set_assoc_grid(1)          # identical records to the brute force (tests/test_oracle_grid.py), just faster
orc = OracleBackend(opts)
dd = sliding.SlidingWindowDriver(ctx, opts, lidar_pose=capi.lidar_pose)
do = sliding.SlidingWindowDriver(orc, opts, lidar_pose=po.lidar_pose_for_association)
dd.start(first); do.start(first)


def body(j):
    c = long.scans[j].copy(); c[:, :3] -= tlb
    return c


pushed = 0
rows = []
tot = dict(kept_dev=0, kept_orc=0, only_dev=0, only_orc=0, common=0, common_differing_bits=0)
for k in range(L - W + 1):
    while pushed < k + W:
        ctx.localmap_push(body(pushed), long.gt.quat[pushed], long.gt.trans[pushed]); pushed += 1
    ctx.localmap_build()
    map_dev = ctx.localmap_read()
    lo = max(0, pushed - 50)
    cloud = np.concatenate([po.transform_cloud(body(j), long.gt.quat[j], long.gt.trans[j]) for j in range(lo, pushed)])
    map_orc, _ = po.voxel_grid(cloud, 0.4)
    sd, smd, cd = dd.step(map_dev, long.scans[k:k + W], long.preints[k:k + W - 1])
    so, smo, co = do.step(map_orc, long.scans[k:k + W], long.preints[k:k + W - 1])
    row = dict(keyframe=k, map_points=[int(len(map_dev)), int(len(map_orc))], same_voxels=bool(len(map_dev) == len(map_orc)),
               centroid_max_diff_m=float(np.abs(map_dev[:, :3] - map_orc[:, :3]).max()) if len(map_dev) == len(map_orc) else None,
               iterations=[int(smd.iterations), int(smo.iterations)], max_trans_diff_m=float(np.linalg.norm(sd.trans - so.trans, axis=1).max()),
               only_dev=0, only_orc=0, common_differing_bits=0, kept=[int(sum(cd)), int(sum(co))])
    for s in range(W):
        pd, pld, scd = ctx.get_correspondences(s)
        po_, plo, sco = orc.corr[s]
        kd = {bytes(p): i for i, p in enumerate(np.ascontiguousarray(pd[:, :3]))}
        ko = {bytes(p): i for i, p in enumerate(np.ascontiguousarray(po_[:, :3]))}
        common = kd.keys() & ko.keys()
        row["only_dev"] += len(kd.keys() - ko.keys()); row["only_orc"] += len(ko.keys() - kd.keys())
        for key in common:
            a, b = kd[key], ko[key]
            if pld[a].tobytes() != plo[b].tobytes() or scd[a] != sco[b]:
                row["common_differing_bits"] += 1
        tot["common"] += len(common)
    tot["kept_dev"] += row["kept"][0]; tot["kept_orc"] += row["kept"][1]
    for key in ("only_dev", "only_orc", "common_differing_bits"):
        tot[key] += row[key]
    rows.append(row)
    if k + W < L:
        for d in (dd, do):
            d.slide(long.init.trans[k + W], long.init.quat[k + W], long.init.speed_bias[k + W])
po.lib().orc_set_assoc_grid(0)
ctx.close()
out = dict(what="device-built map + prior chain vs oracle-built, same stream; association records compared per keyframe", localmap_accumulation=ACC, keyframes=len(rows), window=W, points_per_scan=pts,
           totals=tot, fraction_of_records_in_one_side_only=(tot["only_dev"] + tot["only_orc"]) / max(1, tot["kept_dev"] + tot["kept_orc"]),
           fraction_of_common_records_with_different_bits=tot["common_differing_bits"] / max(1, tot["common"]),
           max_trans_diff_m=max(r["max_trans_diff_m"] for r in rows), per_keyframe=rows)
print(json.dumps(out))
