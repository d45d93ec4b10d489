"""Generates tests/golden/window_small.npz: a small sliding window (fixed seed) pushed through the CPU oracle.

What it pins.  The reference (C++/ROS/Ceres) cannot be built or run in this image and ships no test vectors for this path
(SURVEY.md section 4), so these are NOT reference outputs -- parity stays "unpinned" (oracle/glio_oracle.h).  The file freezes
the oracle's own answers for one small case: tests/test_golden.py checks, without a GPU, that the oracle (and the synthetic
generator behind the inputs) still reproduce them, and, with a GPU, that the HIP path agrees with the frozen numbers, so
that a silent change on either side shows up as a diff against a committed artefact.

    python tests/golden/make_golden.py          # rewrites window_small.npz (review the diff of test results first)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

SEED_OFFSET = 77
W, PTS = 3, 384


def make_case():
    from glio_amd import synth
    from oracle import pyoracle as po
    win = synth.make_window(W=W, pts_per_scan=PTS, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + SEED_OFFSET)
    corr, assoc_counts, assoc_head = [], [], []
    for s in range(win.W):
        q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
        pts, pl, sc, _ = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
        corr.append((pts, pl, sc))
        assoc_counts.append(len(sc))
        assoc_head.append(np.concatenate([pts[:8].ravel(), pl[:8].ravel(), sc[:8]]))
    return win, corr, np.array(assoc_counts), np.array(assoc_head)


def input_digest(win):
    h = hashlib.sha256()
    for a in [win.map_pts] + list(win.scans) + [win.init.trans, win.init.quat, win.init.speed_bias, win.init.rcv_ddt]:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def oracle_outputs(win, corr):
    from oracle import pyoracle as po
    prob = po.Problem(win, corr)
    st = win.init.copy()
    H, g, cost = prob.linearize(st)
    sol, summ = prob.solve(st)
    m = prob.marginalize(sol)
    J0 = np.asarray(m["lin_jac"]); r0 = np.asarray(m["lin_res"])
    return dict(H=H, g=g, cost=np.array(cost), sol_trans=sol.trans, sol_quat=sol.quat, sol_speed_bias=sol.speed_bias,
                sol_rcv_ddt=np.asarray(sol.rcv_ddt)[: sol.n_ddt], iterations=np.array(summ.iterations), final_cost=np.array(summ.final_cost),
                marg_S=J0.T @ J0, marg_b=J0.T @ r0, marg_c=np.array(r0 @ r0))


if __name__ == "__main__":
    win, corr, counts, head = make_case()
    out = oracle_outputs(win, corr)
    out.update(assoc_counts=counts, assoc_head=head, input_sha256=np.array(input_digest(win)))
    path = os.path.join(HERE, "window_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; H", out["H"].shape, "iterations", int(out["iterations"]), "kept", counts)
