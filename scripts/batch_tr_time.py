"""Wall time of the batch pose problem's trust-region rounds (glio_batch_solve_tr) at C4 size on one GPU."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glio_amd import batch, ctypes_types as T
K, band, per_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 6, int(sys.argv[2]) if len(sys.argv) > 2 else 32768
gt, init = batch.make_poses(K)
ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, device="cuda:0")
st = batch.BatchStage(K, band, len(ci)); st.set_constraints(ci, cj, cp, nc, score)
odo = gt.copy(); odo[:, :3] += np.random.default_rng(11).normal(0, 0.02, (K, 3))
dd, frame = batch.make_batch_gnss(gt, seed=11)
for rep in range(2):
    t0 = time.perf_counter()
    poses, rounds = batch.solve_batch_rounds(st, init, odo, 3, dd, frame, opts=T.batch_tr_opts(10))
    dt = time.perf_counter() - t0
    lins = sum(r["iterations"] + 1 for r in rounds)
    print("wall ms", round(dt * 1e3, 1), "solve ms", [round(r["solve_ms"], 2) for r in rounds], "iterations", [r["iterations"] for r in rounds],
          "ms per linearisation", round(sum(r["solve_ms"] for r in rounds) / lins, 3), "err", np.abs(poses[:, :3] - gt[:, :3]).max())
