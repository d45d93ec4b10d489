"""Times the trust-region solve of the batch problem (pose-only and with the IMU chain) at the C4 shape on one GPU and prints the
per-group time; with rocprofv3 --kernel-trace --stats around it this gives the per-kernel breakdown of a group.
    python scripts/batch_tr_time.py [K] [per_kf]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from glio_amd import batch  # noqa: E402
from glio_amd import ctypes_types as T  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
per_kf = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
band = 6
gt, init = batch.make_poses(K)
ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, device="cuda:0")
st = batch.BatchStage(K, band, len(ci))
st.set_constraints(ci, cj, cp, nc, score)
odo = gt.copy(); odo[:, :3] += np.random.default_rng(11).normal(0, 0.02, (K, 3))
dd, frame = batch.make_batch_gnss(gt, seed=11)
st.set_small_factors(batch.delta_q_pairs(odo, 3), dd, frame, threshold=10.0)
out = {}
for label, with_imu in (("pose_6", False), ("full_15", True)):
    sb0 = None
    if with_imu:
        imu, sb_gt, sb0 = batch.make_batch_imu(K, seed=11)
        st.set_imu(imu)
    opts = T.batch_tr_opts(max_iterations=8)
    st.solve_tr(init, opts, speed_bias=sb0)
    st.counters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = st.solve_tr(init, opts, speed_bias=sb0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = st.counters()
    out[label] = {"solve_ms": round(dt * 1e3, 3), "groups": int(c["groups"]), "ms_per_group": round(dt * 1e3 / c["groups"], 3), "iterations": int(res[-1].iterations),
                  "final_cost": float(res[-1].final_cost)}
print(json.dumps(out))
try:
    import ctypes as C
    from glio_amd import capi
    st8 = (C.c_longlong * 8)()
    if capi.load().glio_debug_bcr_stamps(st8) == 0:
        print("elim2 workgroup 0 of the last launch (us): load %.2f, register steps %.2f, MFMA updates %.2f, store %.2f" % tuple(v / 100.0 for v in list(st8)[:4]))
        print("k_bcr_pre workgroup 0 of the last launch (us): gather %.2f, 36 register steps %.2f, factors out + Schur complement %.2f" % tuple(v / 100.0 for v in list(st8)[4:7]))
except AttributeError:
    pass
