"""Generates tests/golden/ref_factors.npz: outputs of THE REFERENCE'S OWN factor code (oracle/_ref/libglio_ref.so = the reference's
headers + MarginalizationFactor.cpp + gnss_utility.cpp compiled unmodified from /root/reference, see oracle/ref_shim/Makefile) on
seeded random inputs.  /root/reference does not exist on the GPU box, so these vectors are how the reference travels there:
tests/test_golden_ref.py checks the oracle against them on any CPU and the HIP evaluators (glio_eval_*, glio_marginalize) against
them on the GPU.

    python tests/golden/make_golden_ref.py        # needs /root/reference (this container); rewrites ref_factors.npz

Every array named *_in is an input, *_out an output of the reference; structs travel as raw bytes of the ctypes mirror of
include/glio_types.h."""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

N = 64
SEED = 20260926


def rand_q(rng, unit=True):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return q if unit else q * (1.0 + rng.normal() * 1e-3)


def struct_bytes(s):
    return np.frombuffer(bytes(s), np.uint8).copy()


def random_preint(rng):
    from glio_amd import synth
    n = int(rng.integers(5, 45))
    acc = rng.normal(0, 0.8, (n + 1, 3)) + np.array([0, 0, 9.8])
    gyr = rng.normal(0, 0.3, (n + 1, 3))
    dts = rng.uniform(0.004, 0.012, n)
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)
    return synth.preintegrate(acc, gyr, dts, ba, bg), ba, bg


def canonical_marg(out):
    """(J0^T J0, J0^T r0, |r0|^2) in the block order T1 Q1 SB1 T2 Q2 ... (slots after the shift: 0, 1, ...)"""
    from glio_amd import ctypes_types as T
    nb = len(out["blk_slot"])
    order = sorted(range(nb), key=lambda b: (out["blk_slot"][b], out["blk_kind"][b]))
    idx = []
    for b in order:
        sz = 9 if out["blk_kind"][b] == T.BLK_SPEEDBIAS else 3
        idx += list(range(out["blk_idx"][b], out["blk_idx"][b] + sz))
    J = out["lin_jac"][:, idx]
    return J.T @ J, J.T @ out["lin_res"], float(out["lin_res"] @ out["lin_res"]), np.array([(out["blk_slot"][b], out["blk_kind"][b]) for b in order], np.int32), \
        np.array([out["blk_x0"][b] for b in order])


def generate():
    from glio_amd import ctypes_types as T
    from glio_amd import synth
    from oracle import pyoracle as po
    from oracle import pyref as ref
    ref.build()
    rng = np.random.default_rng(SEED)
    g = {}
    # ---- LidarPlaneNormFactor
    a = dict(qlb=[], tlb=[], t=[], q=[], cp=[], plane=[], score=[], r=[], Jt=[], Jq=[])
    extr = [(np.array([1.0, 0, 0, 0]), np.array([0, 0, 0.28]))] + [(rand_q(rng), rng.normal(size=3) * 0.3) for _ in range(3)]     # the yaml's extrinsic + three random ones
    for k in range(N):
        qlb, tlb = extr[k % 4]
        t, q = rng.normal(size=3) * 20, rand_q(rng, k % 5 != 0)
        cp = (rng.normal(size=4) * 15).astype(np.float32)
        w = rng.uniform(0.3, 1.0)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        plane = np.r_[w * n, w * rng.normal() * 10].astype(np.float32)
        score = 7.5 * w
        r, J = ref.eval_lidar_plane(cp[:3].astype(float), plane[:3].astype(float), float(plane[3]), score, qlb, tlb, t, q)
        for key, v in zip(a, (qlb, tlb, t, q, cp, plane, score, r[0], J[0][0], J[1][0])):
            a[key].append(v)
    for key, v in a.items():
        g["lidar_" + key + ("_out" if key in ("r", "Jt", "Jq") else "_in")] = np.array(v)
    # ---- BinaryLidarPlaneNormFactor
    a = dict(t1=[], q1=[], t2=[], q2=[], cp=[], pnc=[], score=[], r=[], J0=[], J1=[], J2=[], J3=[])
    for k in range(N):
        t1, q1, t2, q2 = rng.normal(size=3) * 20, rand_q(rng, k % 4 != 0), rng.normal(size=3) * 20, rand_q(rng, k % 5 != 0)
        cp = (rng.normal(size=4) * 15).astype(np.float32)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        pnc = np.r_[n, rng.normal(size=3) * 15]
        score = 2.5 * rng.uniform(0.3, 1.0)
        r, J = ref.eval_binary_plane(cp[:3].astype(float), pnc, score, t1, q1, t2, q2)
        for key, v in zip(a, (t1, q1, t2, q2, cp, pnc, score, r[0], J[0][0], J[1][0], J[2][0], J[3][0])):
            a[key].append(v)
    for key, v in a.items():
        g["binary_" + key + ("_out" if key[0] in "rJ" else "_in")] = np.array(v)
    # ---- delta_q_factor_auto, LidarPoseFactorBatchRelativeAutoDiff
    a = dict(dq=[], qi=[], qj=[], r=[], J0=[], J1=[])
    for k in range(N):
        dq, qi, qj = rand_q(rng), rand_q(rng, k % 3 != 0), rand_q(rng, k % 4 != 0)
        r, J = ref.eval_delta_q(dq, qi, qj)
        for key, v in zip(a, (dq, qi, qj, r, J[0], J[1])):
            a[key].append(v)
    for key, v in a.items():
        g["deltaq_" + key + ("_out" if key[0] in "rJ" else "_in")] = np.array(v)
    a = dict(dq=[], dp=[], p1=[], q1=[], p2=[], q2=[], r=[], J0=[], J1=[], J2=[], J3=[])
    for k in range(N):
        dq, dp = rand_q(rng), rng.normal(size=3) * 3
        p1, q1, p2, q2 = rng.normal(size=3) * 20, rand_q(rng, k % 3 != 0), rng.normal(size=3) * 20, rand_q(rng, k % 4 != 0)
        r, J = ref.eval_relative_pose(dq, dp, p1, q1, p2, q2)
        for key, v in zip(a, (dq, dp, p1, q1, p2, q2, r, J[0], J[1], J[2], J[3])):
            a[key].append(v)
    for key, v in a.items():
        g["relpose_" + key + ("_out" if key[0] in "rJ" else "_in")] = np.array(v)
    # ---- ImuFactor
    o = synth.default_opts()
    pres, pars, rs, Js = [], [], [], [[] for _ in range(6)]
    for k in range(N):
        pre, ba, bg = random_preint(rng)
        ps = T.GlioPreint(); synth.fill_preint(ps, pre)
        params = [rng.normal(size=3) * 10, rand_q(rng, k % 4 != 0), np.r_[rng.normal(size=3) * 5, ba + rng.normal(0, 0.01, 3), bg + rng.normal(0, 0.003, 3)],
                  rng.normal(size=3) * 10, rand_q(rng, k % 5 != 0), np.r_[rng.normal(size=3) * 5, rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)]]
        r, J = ref.eval_imu(ps, o.gravity, params)
        pres.append(struct_bytes(ps)); pars.append(np.concatenate(params)); rs.append(r)
        for b in range(6):
            Js[b].append(J[b])
    g["imu_preint_in"], g["imu_params_in"], g["imu_gravity_in"], g["imu_r_out"] = np.array(pres), np.array(pars), np.array(o.gravity), np.array(rs)
    for b in range(6):
        g["imu_J%d_out" % b] = np.array(Js[b])
    # ---- Preintegration::push_back (one long sequence; inputs stored)
    for name, v in zip(("/IMU/acc_n", "/IMU/gyr_n", "/IMU/acc_w", "/IMU/gyr_w"), (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)):
        ref.set_param(name, v)
    n = 40
    acc = rng.normal(0, 0.8, (n + 1, 3)) + np.array([0, 0, 9.8]); gyr = rng.normal(0, 0.3, (n + 1, 3)); dts = rng.uniform(0.004, 0.012, n)
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)
    got = ref.preintegrate(acc[0], gyr[0], ba, bg, dts, acc[1:], gyr[1:])
    g["preint_acc_in"], g["preint_gyr_in"], g["preint_dt_in"], g["preint_ba_in"], g["preint_bg_in"], g["preint_out"] = acc, gyr, dts, ba, bg, struct_bytes(got)
    # ---- dd_psr_factor_20, tcdopplerFactor
    win = synth.make_window(W=6, pts_per_scan=64, with_gnss=True, seed=synth.SEED_BASE + 77)
    anc0 = np.array(win.frame.anc_ecef)
    fs, Pis, Pjs, yaws, ancs, rs, J0s, J1s = [], [], [], [], [], [], [], []
    for k in range(N):
        f = copy.copy(win.dd[k % len(win.dd)])
        ns = f.n_sat
        if k % 3 == 0:
            Wm = np.eye(ns - 1) + rng.normal(0, 0.2, (ns - 1, ns - 1))
            f.weight[:(ns - 1) ** 2] = list(Wm.ravel())
        f.threshold = [1e9, 5.0, 0.5, 0.0][k % 4]
        f.ratio = rng.uniform(0, 1)
        Pi, Pj = win.init.trans[f.slot_i] + rng.normal(0, 3, 3), win.init.trans[f.slot_j] + rng.normal(0, 3, 3)
        yaw, anc = rng.uniform(-3, 3), anc0 + rng.normal(0, 50, 3)
        r, J = ref.eval_dd_psr(f, Pi, Pj, yaw, anc)
        fs.append(struct_bytes(f)); Pis.append(Pi); Pjs.append(Pj); yaws.append(yaw); ancs.append(anc); rs.append(r); J0s.append(J[0]); J1s.append(J[1])
    g.update(dd_f_in=np.array(fs), dd_Pi_in=np.array(Pis), dd_Pj_in=np.array(Pjs), dd_yaw_in=np.array(yaws), dd_anc_in=np.array(ancs),
             dd_r_out=np.array(rs), dd_J0_out=np.array(J0s), dd_J1_out=np.array(J1s))
    st = win.init
    nslot = ref.ddt_slots()
    fs, args_all, yaws, ancs, rs, Js = [], [], [], [], [], [[] for _ in range(5)]
    for k in range(N):
        f = copy.copy(win.dop[k % len(win.dop)])
        f.epoch = int(rng.integers(0, nslot)); f.ratio = rng.uniform(0, 1)
        args = [st.trans[f.slot_i] + rng.normal(0, 2, 3), st.speed_bias[f.slot_i] + rng.normal(0, 1, 9), st.trans[f.slot_j] + rng.normal(0, 2, 3),
                st.speed_bias[f.slot_j] + rng.normal(0, 1, 9), rng.normal(0, 3, nslot)]
        yaw, anc = rng.uniform(-3, 3), anc0 + rng.normal(0, 30, 3)
        r, J = ref.eval_doppler(f, *args, yaw, anc)
        fs.append(struct_bytes(f)); args_all.append(np.concatenate(args)); yaws.append(yaw); ancs.append(anc); rs.append(r)
        for b in range(4):
            Js[b].append(J[b][0])
        Js[4].append(J[4][0][f.epoch])
    g.update(dop_f_in=np.array(fs), dop_args_in=np.array(args_all), dop_yaw_in=np.array(yaws), dop_anc_in=np.array(ancs), dop_nslot_in=np.array(nslot), dop_r_out=np.array(rs))
    for b in range(5):
        g["dop_J%d_out" % b] = np.array(Js[b])
    # ---- ecef2rotation
    E = []
    for k in range(N):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        E.append(d * rng.uniform(6.3e6, 6.5e6))
    g["ecef_in"] = np.array(E)
    g["ecef_R_out"] = np.array([ref.ecef2rotation(e) for e in E])
    # ---- MarginalizationFactor::Evaluate on the synthetic prior of a W = 4 window (regenerated from its seed by the tests)
    winp = synth.make_window(W=4, pts_per_scan=32, with_prior=True, seed=synth.SEED_BASE + 123)
    pr = winp.prior
    P, R, JJ = [], [], []
    for k in range(N // 2):
        params = []
        for b in range(len(pr["blk_slot"])):
            kind, x0 = pr["blk_kind"][b], pr["blk_x0"][b]
            if kind == T.BLK_QUAT:
                q = po.quat_plus(x0[:4], rng.normal(size=3) * [1e-3, 0.3, 2.5][k % 3])
                params.append(-q if k % 2 else q)
            else:
                sz = 3 if kind == T.BLK_TRANS else 9
                params.append(x0[:sz] + rng.normal(size=sz) * 0.1)
        r, J = ref.eval_marg(pr, params)
        P.append(np.concatenate(params)); R.append(r); JJ.append(np.concatenate([j.ravel() for j in J]))
    g.update(margf_seed_in=np.array(synth.SEED_BASE + 123), margf_params_in=np.array(P), margf_r_out=np.array(R), margf_J_out=np.array(JJ))
    # ---- the marginalization step on a W = 4 window with a prior (window regenerated from its seed; state and correspondences stored)
    winm = synth.make_window(W=4, pts_per_scan=300, with_prior=True, seed=synth.SEED_BASE + 94)
    corr = synth.analytic_correspondences(winm)
    prob = po.Problem(winm, corr, use_gnss=False, use_prior=True)
    stt = winm.init.copy(); stt.n_ddt = 0
    sol, _ = prob.solve(stt)
    out = ref.marginalize(winm.opts, sol, prob.offset, prob.pts, prob.planes, prob.scores, prob.imu[0], winm.prior)
    S, b, c, order, x0 = canonical_marg(out)
    g.update(marg_seed_in=np.array(synth.SEED_BASE + 94), marg_trans_in=sol.trans, marg_quat_in=sol.quat, marg_sb_in=sol.speed_bias,
             marg_S_out=S, marg_b_out=b, marg_c_out=np.array(c), marg_order_out=order, marg_x0_out=x0)
    return g


if __name__ == "__main__":
    g = generate()
    path = os.path.join(HERE, "ref_factors.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path), "bytes,", len(g), "arrays")
