"""Where the association of all pairs of the C4-sized batch stage spends its time: the three runs and the feed of RoundsAssociation.start, timed apart."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glio_amd import batch, synth
K = int(os.environ.get("PK", "2000")); pts = int(os.environ.get("PPTS", "32768")); sr = 6; distinct = 32
win = synth.make_window(W=distinct, pts_per_scan=pts, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=25.0, map_density=0.5)
tlb = np.array(win.opts.t_lb, np.float32)
base = []
for k in range(distinct):
    sc = win.scans[k].copy(); sc[:, :3] -= tlb
    base.append(np.ascontiguousarray(sc))
period = 2 * (distinct - 1)
tri = [(k % period) if (k % period) < distinct else period - (k % period) for k in range(K)]
scans = [base[i] for i in tri]
poses = np.c_[win.init.trans, win.init.quat][tri]
ci, cj = batch.pair_list(K, sr)
st = batch.BatchStage(K, 2 * sr, int(len(ci)) * pts, device=0)
ra = batch.RoundsAssociation(st, scans, sr, pts, device=0)
for rep in range(3):
    ts = []
    for w in range(3):
        t0 = time.perf_counter(); ra._run(w, poses); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ra._feed(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    print("rep", rep, "runs ms", [round(t * 1e3, 2) for t in ts], "pairs", [len(p[1]) for p in ra.parts], "feed ms", round(tf * 1e3, 2), "constraints", ra.n_constraints, flush=True)
