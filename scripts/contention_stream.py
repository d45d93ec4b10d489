"""N processes sharing one GPU; each runs the resident keyframe stream (slide, new scan, local-map push of the resident scan + build, asynchronous association
of all slots with the factor tables staged meanwhile, solve, marginalize-and-keep) over a few keyframes, TWICE on fresh contexts, and compares the two
runs bit for bit (poses, iteration counts, correspondence counts, map sizes).  Determinism under load = no unordered operation left in that path."""
import os, subprocess, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("REPRO_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = sys.argv[2] if len(sys.argv) > 2 else "3"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(os.environ, REPRO_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi
from glio_amd.capi import lidar_pose
rank = int(os.environ["REPRO_RANK"]); reps = int(sys.argv[1])
W, pts, nkf = 8, 16384, 5
long = synth.make_window(W=W + nkf, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 300 + rank)
wins = [synth.sub_window(long, j, W) for j in range(nkf + 1)]
tlb = np.array(wins[0].opts.t_lb, np.float32)


def run():
    opts = wins[0].opts
    opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
    opts.max_map_points = 1 << 18
    ctx = capi.Context(opts); ctx.localmap_config(50, 0.4, pts)
    for j in range(W - 1):
        c = long.scans[j].copy(); c[:, :3] -= tlb
        ctx.localmap_push(np.ascontiguousarray(c), long.gt.quat[j], long.gt.trans[j])
    for s in range(W - 1):
        ctx.set_scan(s + 1, long.scans[s])
    ctx.set_prior(None)
    state = wins[0].init.copy()
    trace = []
    sol = None
    for j in range(nkf + 1):
        win = wins[j]; new = j + W - 1
        if j > 0:
            nxt = win.init.copy()
            nxt.trans[:-1], nxt.quat[:-1], nxt.speed_bias[:-1] = sol.trans[1:], sol.quat[1:], sol.speed_bias[1:]
            state = nxt
        ctx.slide_window(); ctx.set_scan(W - 1, long.scans[new])
        ctx.localmap_push_scan(W - 1, tlb, long.gt.quat[new], long.gt.trans[new])
        nmap = ctx.localmap_build()
        poses = [lidar_pose(opts, state.quat[s], state.trans[s]) for s in range(W)]
        ctx.associate_window_async(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
        ctx.set_imu(win.preints); ctx.set_gnss(win.frame, win.dd, win.dop)
        counts = ctx.associate_window_counts()
        sol, summ = ctx.solve(state)
        ctx.marginalize_keep(sol)
        trace.append((int(nmap), counts.tobytes(), sol.trans.tobytes(), sol.quat.tobytes(), sol.speed_bias.tobytes(), int(summ.iterations), int(summ.termination)))
    ctx.close()
    return trace


events = []
try:
    ref = run()
    for it in range(reps):
        got = run()
        for k, (a, b) in enumerate(zip(ref, got)):
            if a != b:
                events.append([it, k, [i for i in range(len(a)) if a[i] != b[i]]]); break
except Exception as e:
    events.append(["error", str(e)[-100:]])
print(json.dumps({"rank": rank, "events": events[:4], "keyframes": nkf + 1, "iterations": [t[5] for t in ref] if 'ref' in dir() else None}))
