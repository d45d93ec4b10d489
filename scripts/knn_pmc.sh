#!/bin/bash
# hardware counters of k_knn5 (one --pmc pass per group): what bounds the association kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/knn_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from glio_amd import synth, capi
from glio_amd.capi import lidar_pose
WIN = bool(os.environ.get("KNN_WINDOW"))
win = synth.make_window(W=20 if WIN else 2, pts_per_scan=65536, seed=synth.SEED_BASE if WIN else synth.SEED_BASE + 12)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
for s in range(win.W): ctx.set_scan(s, win.scans[s])
q, t = lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
if WIN:
    poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
    q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
    for _ in range(3): ctx.associate_window(q2s, t2s)
else:
    for _ in range(4): ctx.associate_resident(0, q, t)
if os.environ.get("KNN_UNITS"):
    R = synth.q2R(np.asarray(q)); p = (win.scans[0][:, :3].astype(np.float64) @ R.T + np.asarray(t)).astype(np.float32)
    cell = np.floor(p * np.float32(1.0 / 1.25)).astype(np.int64)
    _, cnt = np.unique(cell, axis=0, return_counts=True)
    print("query cells", len(cnt), "units", int(np.sum((cnt + 15) // 16)), "mean fill", float(np.mean(cnt)), "median", float(np.median(cnt)), "max", int(cnt.max()))
PY
KNN_UNITS=1 python /tmp/knn_once.py | tail -1
for GRP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT"; do
  OUT=/tmp/knn_pmc; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $GRP --output-format csv -d $OUT -o pmc -- python /tmp/knn_once.py > /tmp/knn_pmc.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        kn = row["Kernel_Name"].split("(")[0]
        if "k_knn5" in kn or "k_qbin" in kn or "k_gbin" in kn or "k_plane" in kn:
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn:16s} {k:32s} last launch {v[-1]:16.1f}  (launches {len(v)})")
    if not acc: print("no k_knn5 rows:", open("/tmp/knn_pmc.log").read()[-500:])
except Exception as e:
    print("no data:", e); print(open("/tmp/knn_pmc.log").read()[-600:])
PY
done
