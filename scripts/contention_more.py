"""N processes sharing one GPU; each repeats, on FRESH objects: (a) K1 + K2 association of a scan, (b) the batch problem with the IMU chain
(trust-region solve), (c) a local-map build -- and compares every result with its own first one (hash of the outputs)."""
import os, subprocess, sys, json, hashlib
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("REPRO_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = sys.argv[2] if len(sys.argv) > 2 else "10"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(os.environ, REPRO_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi, batch
from glio_amd import ctypes_types as T
from glio_amd.capi import lidar_pose
rank = int(os.environ["REPRO_RANK"]); reps = int(sys.argv[1])
win = synth.make_window(W=2, pts_per_scan=32768, seed=synth.SEED_BASE + 100 + rank)
q2, t2 = lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
K, band = 36, 6
gt, init = batch.make_poses(K, seed=60 + rank, perturb=(0.08, 0.004))
ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 80, band, seed=60 + rank)
con = (ci, cj, cp.numpy(), nc.numpy(), score.numpy())
dq = batch.delta_q_pairs(gt, 3); dd, frame = batch.make_batch_gnss(gt, seed=60 + rank)
for f in dd: f.threshold = 10.0
imu, _, sb0 = batch.make_batch_imu(K, seed=60 + rank)
def h(*arrs):
    m = hashlib.sha1()
    for a in arrs: m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:12]
seen, events = {}, []
for it in range(reps):
    try:
        c = capi.Context(win.opts); c.set_map(win.map_pts)
        n = c.associate(0, win.scans[0], q2, t2); rec = c.get_correspondences(0)
        c.localmap_config(8, 0.4, 32768)
        for j in range(2): c.localmap_push(win.scans[j], win.init.quat[j], win.init.trans[j])
        nm = c.localmap_build(); lm = c.localmap_read()
        c.close()
        st = batch.BatchStage(K, band, len(ci)); st.set_constraints(*con); st.set_small_factors(dq, dd, frame); st.set_imu(imu)
        poses, sb, sm = st.solve_tr(init, T.batch_tr_opts(max_iterations=8), speed_bias=sb0); st.close()
        key = (int(n), h(*rec), int(nm), h(lm), sm.iterations, sm.termination, h(poses, sb))
        seen[key] = seen.get(key, 0) + 1
        if len(seen) > 1 and seen[key] == 1: events.append([it, list(key)])
    except Exception as e:
        events.append([it, str(e)[-80:]])
print(json.dumps({"rank": rank, "distinct": len(seen), "events": events[:5], "first": list(list(seen.keys())[0]) if seen else None}))
