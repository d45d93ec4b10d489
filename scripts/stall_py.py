"""The slow marginalization of NOTES_r05.md (Python stream driver, batch association PREPARED before the solve, seconds of host work between the solve and the
association's enqueue: the marginalization stage then read 11-25 ms).  Same driver, with the host activity of the gap selectable: STALL_GAP = none | sleep |
churn (allocate / touch / free 2 GB of numpy arrays, twice) | oracle (one CPU solve of the same window through oracle/) | spin (busy loop).  Prints per keyframe
the marginalize stage and the stage that follows."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import capi, synth, batch, sliding
from glio_amd.capi import lidar_pose
W, pts, NK = 20, int(os.environ.get("STALL_PTS", "65536")), 6
gap = os.environ.get("STALL_GAP", "none"); prep = os.environ.get("STALL_PREPARE", "1") == "1"; gap_s = float(os.environ.get("STALL_GAP_S", "2.0"))
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
ctx = capi.Context(opts)
ctx.localmap_config(50, 0.4, pts)
tlb = np.array(opts.t_lb, np.float32)
bodies = []
for j in range(W + NK):
    c = long.scans[j].copy(); c[:, :3] -= tlb
    bodies.append(np.ascontiguousarray(c))
for j in range(W - 1):
    ctx.localmap_push(bodies[j], long.gt.quat[j], long.gt.trans[j])
for s in range(W - 1):
    ctx.set_scan(s + 1, long.scans[s])
ctx.set_prior(None)
total_kf = W + NK
ba = batch.BatchAssociation(total_kf, pts, (NK + 2) * 12 * pts)
kba = sliding.KeyframeBatchAssociation(ba, search_range=6, feature_res_num=25, rng=np.random.default_rng(1))
kf_poses = np.c_[long.gt.trans, long.gt.quat][:total_kf].copy()
for j in range(W - 1):
    ba.set_frame(j, bodies[j])
prob = None
if gap == "oracle":
    from oracle import pyoracle as po
sol = None
rows = []
for j in range(NK + 1):
    win = wins[j]
    state = win.init.copy()
    if j > 0:
        state.trans[:-1], state.quat[:-1], state.speed_bias[:-1] = sol.trans[1:], sol.quat[1:], sol.speed_bias[1:]
    new = j + W - 1
    ctx.slide_window(); ctx.set_scan(W - 1, long.scans[new])
    ctx.localmap_push_scan(W - 1, tlb, long.gt.quat[new], long.gt.trans[new]); ctx.localmap_build()
    poses = [lidar_pose(opts, state.quat[s], state.trans[s]) for s in range(W)]
    ctx.associate_window_async(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
    ctx.set_imu(win.preints); ctx.set_gnss(win.frame, win.dd, win.dop)
    counts = ctx.associate_window_counts()
    if prep:
        kba.prepare(new + 1)
    t4 = time.perf_counter(); sol, summ = ctx.solve(state); t5 = time.perf_counter()
    if j >= 2 and gap != "none":
        g0 = time.perf_counter()
        if gap == "sleep":
            time.sleep(gap_s)
        elif gap == "spin":
            while time.perf_counter() - g0 < gap_s:
                pass
        elif gap == "churn":
            for _ in range(2):
                a = np.ones(1 << 28); a += 1.0; del a
        elif gap == "oracle":
            corr = [ctx.get_correspondences(s) for s in range(W)]
            pw = synth.sub_window(long, j, W); pw.prior = None
            po.Problem(pw, corr, use_prior=False).solve(state)
    usol = sliding.unify_quaternions(sol.copy())
    kf_poses[j:j + W, :3] = usol.trans; kf_poses[j:j + W, 3:] = usol.quat
    t5c = time.perf_counter()
    ba.set_frame_from_scan(new, ctx, W - 1, tlb)
    kba.enqueue(new + 1, kf_poses)
    t6 = time.perf_counter(); ctx.marginalize_keep(sol); t7 = time.perf_counter()
    found = kba.finish(); t8 = time.perf_counter()
    rows.append({"j": j, "solve_ms": round((t5 - t4) * 1e3, 3), "enqueue_ms": round((t6 - t5c) * 1e3, 3), "marginalize_ms": round((t7 - t6) * 1e3, 3), "finish_ms": round((t8 - t7) * 1e3, 3)})
print(json.dumps({"gap": gap, "prepare": prep, "rows": rows}))
