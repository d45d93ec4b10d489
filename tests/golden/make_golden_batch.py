"""Generates tests/golden/batch_small.npz: frozen ORACLE answers (not reference output -- see make_golden.py) for the batch pose
problem of optimizeBatch (SURVEY section 8 row a13 + the small factors): a 16-keyframe batch with pre-associated binary plane
constraints, the delta_q attitude constraints of the reference's walk and one DD-pseudorange factor pair per keyframe gap; the
linearisation at the initial poses and the four DDpsr_threshold rounds of the trust-region solve.

    python tests/golden/make_golden_batch.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

K, BAND, PER_KF, SEARCH_RANGE, MAX_ITER = 16, 6, 40, 3, 25


def make_inputs():
    from glio_amd import batch
    gt, init = batch.make_poses(K, seed=20260931, perturb=(0.06, 0.004))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, PER_KF, BAND, seed=20260931)
    odo = gt.copy()
    odo[:, :3] += np.random.default_rng(20260931).normal(0, 0.02, (K, 3))
    dd, frame = batch.make_batch_gnss(gt, seed=20260931, sats_per_sys=7)
    return dict(gt=gt, init=init, con=(ci, cj, cp.numpy(), nc.numpy(), score.numpy()), odo=odo, dd=dd, frame=frame)


def digest(case):
    h = hashlib.sha256()
    for a in (case["gt"], case["init"], case["odo"]) + tuple(case["con"]):
        h.update(np.ascontiguousarray(a).tobytes())
    for f in case["dd"]:
        thr, f.threshold = f.threshold, 0.0          # (the rounds overwrite the threshold: not part of the input)
        h.update(bytes(f))
        f.threshold = thr
    return h.hexdigest()


def oracle_outputs(case):
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    from oracle import pyoracle as po
    dq = batch.delta_q_pairs(case["odo"], SEARCH_RANGE)
    for f in case["dd"]:
        f.threshold = batch.DDPSR_THRESHOLDS[0]
    H, g, cost = po.BatchProblem(K, BAND, *case["con"], dq=dq, dd=case["dd"], frame=case["frame"]).linearize(case["init"])
    poses = case["init"].copy()
    its, terms, costs = [], [], []
    for thr in batch.DDPSR_THRESHOLDS:
        for f in case["dd"]:
            f.threshold = thr
        poses, summ = po.BatchProblem(K, BAND, *case["con"], dq=dq, dd=case["dd"], frame=case["frame"]).solve(poses, T.batch_tr_opts(MAX_ITER))
        its.append(summ.iterations); terms.append(summ.termination); costs.append([summ.initial_cost, summ.final_cost])
    return dict(dq_i=dq[0], dq_j=dq[1], dq_const=dq[2], lin_H=H, lin_g=g, lin_cost=np.float64(cost), round_iterations=np.array(its),
                round_termination=np.array(terms), round_costs=np.array(costs), poses=poses)


def imu_inputs():
    """the same batch with the ImuFactor chain: pre-integrations of the analytic track + perturbed speed-bias blocks"""
    from glio_amd import batch
    imu, sb_gt, sb0 = batch.make_batch_imu(K, seed=20260931)
    return imu, sb0


def imu_digest(imu, sb0):
    h = hashlib.sha256()
    for d in imu:
        for k in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
            h.update(np.ascontiguousarray(d[k], np.float64).tobytes())
    h.update(np.ascontiguousarray(sb0).tobytes())
    return h.hexdigest()


def oracle_outputs_imu(case, imu, sb0):
    """15 unknowns per keyframe, SUBSPACE_DOGLEG, non-monotonic steps, one solve at threshold 10 from a small initial radius (so
    that boundary-constrained subspace steps occur)"""
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    from oracle import pyoracle as po
    dq = batch.delta_q_pairs(case["odo"], SEARCH_RANGE)
    for f in case["dd"]:
        f.threshold = 10.0
    P = po.BatchProblem(K, BAND, *case["con"], dq=dq, dd=case["dd"], frame=case["frame"], imu=imu)
    H, g, cost = P.linearize_dense(case["init"], sb0)
    opts = T.batch_tr_opts(MAX_ITER)
    opts.initial_trust_region_radius = 2.0
    poses, sb, summ, hist = P.solve2(case["init"], opts, sb0, want_history=True)
    return dict(imu_diag=np.diag(H).copy(), imu_g=g, imu_cost=np.float64(cost), imu_iterations=np.int64(summ.iterations), imu_successful=np.int64(summ.successful_steps),
                imu_termination=np.int64(summ.termination), imu_costs=np.array([summ.initial_cost, summ.final_cost]), imu_history=hist, imu_poses=poses, imu_sb=sb)


if __name__ == "__main__":
    case = make_inputs()
    imu, sb0 = imu_inputs()
    out_imu = oracle_outputs_imu(case, imu, sb0)
    np.savez_compressed(os.path.join(HERE, "batch_imu_small.npz"), input_sha256=digest(case), imu_sha256=imu_digest(imu, sb0), **out_imu)
    print("wrote batch_imu_small.npz: iterations", int(out_imu["imu_iterations"]), "termination", int(out_imu["imu_termination"]), "costs", out_imu["imu_costs"])
    out = oracle_outputs(case)
    np.savez_compressed(os.path.join(HERE, "batch_small.npz"), input_sha256=digest(case), **out)
    print("wrote batch_small.npz:", {k: np.shape(v) for k, v in out.items()}, "iterations", out["round_iterations"], "costs", out["round_costs"][:, 1])
