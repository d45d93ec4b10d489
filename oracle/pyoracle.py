"""ctypes binding of the CPU ORACLE (oracle/_build/libglio_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under glio_amd/ may import this module.  Numbers produced with it are
"reference-restatement (Ceres-1.14 semantics)", factor layer pinned on oracle/_ref, solve loop and association unpinned (see oracle/glio_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from glio_amd import ctypes_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libglio_oracle.so")


def build(force=False):
    """make decides what is stale (the Makefile lists the sources and ../include/glio_types.h: a changed struct must rebuild the checker)."""
    try:
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        if force or not os.path.exists(_SO):          # (a prebuilt checker travels with the tree: a box without make / gcc still runs it)
            raise
    return _SO


class OrcProblem(C.Structure):
    _fields_ = [("opts", T.GlioOpts), ("lidar_offset", T.c_int32_p), ("lidar_pts", T.c_float_p),
                ("lidar_planes", T.c_float_p), ("lidar_scores", T.c_double_p),
                ("n_imu", C.c_int32), ("imu", C.POINTER(T.GlioPreint)), ("imu_slot", T.c_int32_p),
                ("prior", T.GlioPrior), ("n_dd", C.c_int32), ("dd", C.POINTER(T.GlioDdPsr)),
                ("n_dop", C.c_int32), ("dop", C.POINTER(T.GlioDoppler)), ("frame", T.GlioGnssFrame)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_associate.restype = C.c_int
        _lib.orc_marginalize.restype = C.c_int
    return _lib


def set_threads(t):
    """OpenMP threads of the LiDAR factor loop (bench.py's all-cores baseline; 1 = parity path)."""
    lib().orc_set_threads.restype = None
    lib().orc_set_threads(int(t))


def _pp(arrs):
    """double const* const* from a list of numpy arrays (or None)."""
    P = (T.c_double_p * len(arrs))()
    for i, a in enumerate(arrs):
        P[i] = T.dptr(a) if a is not None else None
    return P


class Problem:
    """Owns the numpy buffers behind an orc_problem."""

    def __init__(self, win, corr, use_gnss=True, use_prior=True, use_imu=True):
        from glio_amd import synth
        self.win = win
        W = win.W
        counts = [len(c[2]) for c in corr]
        self.offset = np.zeros(W + 1, np.int32)
        self.offset[1:] = np.cumsum(counts)
        self.pts = np.ascontiguousarray(np.vstack([c[0] for c in corr]).astype(np.float32)) if sum(counts) else np.zeros((1, 4), np.float32)
        self.planes = np.ascontiguousarray(np.vstack([c[1] for c in corr]).astype(np.float32)) if sum(counts) else np.zeros((1, 4), np.float32)
        self.scores = np.ascontiguousarray(np.concatenate([c[2] for c in corr])) if sum(counts) else np.zeros(1)
        n_imu = len(win.preints) if use_imu else 0
        self.imu = T.preint_array(n_imu)
        for k in range(n_imu):
            synth.fill_preint(self.imu[k], win.preints[k])
        self.imu_slot = np.arange(max(n_imu, 1), dtype=np.int32)
        self.dd = (T.GlioDdPsr * max(len(win.dd), 1))(*win.dd) if (use_gnss and win.dd) else (T.GlioDdPsr * 1)()
        self.dop = (T.GlioDoppler * max(len(win.dop), 1))(*win.dop) if (use_gnss and win.dop) else (T.GlioDoppler * 1)()
        p = OrcProblem()
        p.opts = win.opts
        p.lidar_offset, p.lidar_pts, p.lidar_planes, p.lidar_scores = T.iptr(self.offset), T.fptr(self.pts), T.fptr(self.planes), T.dptr(self.scores)
        p.n_imu, p.imu, p.imu_slot = n_imu, self.imu, T.iptr(self.imu_slot)
        self.prior_dict = win.prior if use_prior else None
        p.prior = synth.prior_struct(self.prior_dict)
        p.n_dd = len(win.dd) if use_gnss else 0
        p.dd = self.dd
        p.n_dop = len(win.dop) if use_gnss else 0
        p.dop = self.dop
        if win.frame is not None:
            p.frame = win.frame
        self.c = p

    def n(self, state):
        return 15 * self.win.W + state.n_ddt

    def linearize(self, state, want_H=True):
        n = self.n(state)
        H = np.zeros((n, n)) if want_H else None
        g = np.zeros(n) if want_H else None
        cost = C.c_double()
        cs = state.c()
        ok = lib().orc_linearize(C.byref(self.c), C.byref(cs), T.dptr(H) if want_H else None, T.dptr(g) if want_H else None, C.byref(cost))
        assert ok
        return H, g, cost.value

    def solve(self, state):
        s = state.copy()
        cs = s.c()
        summ = T.GlioSummary()
        lib().orc_solve(C.byref(self.c), C.byref(cs), C.byref(summ))
        return s, summ

    def solve_history(self, state):
        """orc_solve_history: (state, summary, rows of (candidate cost, radius of the step, |x - candidate|) per iteration)"""
        s = state.copy()
        cs = s.c()
        summ = T.GlioSummary()
        hist = np.zeros((self.c.opts.max_iterations + 1, 3))
        lib().orc_solve_history(C.byref(self.c), C.byref(cs), C.byref(summ), T.dptr(hist))
        return s, summ, hist[:summ.iterations]

    def marginalize(self, state):
        W = self.win.W
        n = 6 * (W - 1) + 9
        nb = 2 * (W - 1) + 1
        out = dict(n=n, lin_jac=np.zeros((n, n)), lin_res=np.zeros(n), blk_slot=np.zeros(nb, np.int32),
                   blk_kind=np.zeros(nb, np.int32), blk_idx=np.zeros(nb, np.int32), blk_x0=np.zeros((nb, 9)))
        cs = state.c()
        r = lib().orc_marginalize(C.byref(self.c), C.byref(cs), T.dptr(out["lin_jac"]), T.dptr(out["lin_res"]),
                                  T.iptr(out["blk_slot"]), T.iptr(out["blk_kind"]), T.iptr(out["blk_idx"]), T.dptr(out["blk_x0"]))
        assert r == n
        return out


def associate(opts, map_pts, scan, q, t, want_nn=False, threads=1):
    """threads > 1: the brute-force nearest-neighbour phase on that many OpenMP threads (identical output)."""
    n = len(scan)
    pts = np.zeros((n, 4), np.float32)
    planes = np.zeros((n, 4), np.float32)
    scores = np.zeros(n)
    src = np.zeros(n, np.int32)
    nn = np.zeros((n, 5), np.int32) if want_nn else None
    q = np.ascontiguousarray(q, float)
    t = np.ascontiguousarray(t, float)
    lib().orc_associate_mt.restype = C.c_int
    cnt = lib().orc_associate_mt(C.byref(opts), T.fptr(map_pts), len(map_pts), T.fptr(scan), n, T.dptr(q), T.dptr(t),
                                 T.fptr(pts), T.fptr(planes), T.dptr(scores), T.iptr(src), T.iptr(nn) if want_nn else None, int(threads))
    res = (pts[:cnt].copy(), planes[:cnt].copy(), scores[:cnt].copy(), src[:cnt].copy())
    return res + ((nn,) if want_nn else ())


def transform_cloud(pts, q, t):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros_like(pts)
    q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
    lib().orc_transform_cloud.restype = None
    lib().orc_transform_cloud(T.fptr(pts), len(pts), T.dptr(q), T.dptr(t), T.fptr(out))
    return out


def voxel_grid(pts, leaf):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros((max(len(pts), 1), 4), np.float32); idx = np.zeros(max(len(pts), 1), np.int64)
    nv = lib().orc_voxel_grid(T.fptr(pts), len(pts), C.c_float(leaf), T.fptr(out), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:nv].copy(), idx[:nv].copy()


def associate_pair(scan_a, pose_a, scan_b, pose_b):
    """findGlobalCorrespondingSurfFeaturesAdd_Batch for one pair; pose = (t[3], q[4])."""
    na = len(scan_a)
    cp = np.zeros((max(na, 1), 4), np.float32); nc = np.zeros((max(na, 1), 6)); sc = np.zeros(max(na, 1)); src = np.zeros(max(na, 1), np.int32)
    pa = np.ascontiguousarray(pose_a, float); pb = np.ascontiguousarray(pose_b, float)
    qa, ta, qb, tb = pa[3:].copy(), pa[:3].copy(), pb[3:].copy(), pb[:3].copy()
    sa = np.ascontiguousarray(scan_a, np.float32); sb = np.ascontiguousarray(scan_b, np.float32)
    cnt = lib().orc_associate_pair(T.fptr(sa), na, T.dptr(qa), T.dptr(ta), T.fptr(sb), len(sb), T.dptr(qb), T.dptr(tb),
                                   T.fptr(cp), T.dptr(nc), T.dptr(sc), T.iptr(src))
    return cp[:cnt].copy(), nc[:cnt].copy(), sc[:cnt].copy(), src[:cnt].copy()


def lidar_pose_for_association(opts, q, t):
    """Q2 = Q * q_lb^-1, T2 = T - Q2 * t_lb  (Estimator.cpp:2216-2217)."""
    from glio_amd import synth
    qlb = np.array(opts.q_lb)
    q2 = synth.qmul(np.asarray(q, float), synth.qconj(qlb) / (qlb @ qlb))
    t2 = np.asarray(t, float) - synth.q2R(q2 / np.linalg.norm(q2)) @ np.array(opts.t_lb)
    return q2, t2


def eval_lidar_plane(opts, cp, plane, score, t, q, want_J=True):
    r = np.zeros(1)
    Jt, Jq = np.zeros(3), np.zeros(4)
    cp = np.ascontiguousarray(cp, np.float32)
    plane = np.ascontiguousarray(plane, np.float32)
    lib().orc_eval_lidar_plane(C.byref(opts), T.fptr(cp), T.fptr(plane), C.c_double(score), _pp([np.ascontiguousarray(t, float), np.ascontiguousarray(q, float)]),
                               T.dptr(r), _pp([Jt, Jq]) if want_J else None)
    return r[0], Jt, Jq


def eval_imu(opts, pre_struct, params, want_J=True):
    r = np.zeros(15)
    sizes = [3, 4, 9, 3, 4, 9]
    J = [np.zeros((15, s)) for s in sizes]
    params = [np.ascontiguousarray(p, float) for p in params]
    ok = lib().orc_eval_imu(C.byref(opts), C.byref(pre_struct), _pp(params), T.dptr(r), _pp(J) if want_J else None)
    assert ok
    return r, J


def eval_marg(prior_dict, params, want_J=True):
    from glio_amd import synth
    ps = synth.prior_struct(prior_dict)
    n = prior_dict["n"]
    r = np.zeros(n)
    sizes = [3 if k == 0 else (4 if k == 1 else 9) for k in prior_dict["blk_kind"]]
    J = [np.zeros((n, s)) for s in sizes]
    params = [np.ascontiguousarray(p, float) for p in params]
    lib().orc_eval_marg(C.byref(ps), _pp(params), T.dptr(r), _pp(J) if want_J else None)
    return r, J


def eval_dd_psr(f, Pi, Pj, yaw, anc, want_J=True):
    r = np.zeros(19)
    J = [np.zeros((19, 3)), np.zeros((19, 3)), None, None]
    params = [np.ascontiguousarray(Pi, float), np.ascontiguousarray(Pj, float), np.array([yaw], float), np.ascontiguousarray(anc, float)]
    lib().orc_eval_dd_psr(C.byref(f), _pp(params), T.dptr(r), _pp(J) if want_J else None)
    return r, J[:2]


def eval_doppler(f, Pi, SBi, Pj, SBj, ddt, yaw, anc, want_J=True):
    r = np.zeros(1)
    J = [np.zeros(3), np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(1), None, None]
    params = [np.ascontiguousarray(a, float) for a in (Pi, SBi, Pj, SBj, ddt)] + [np.array([yaw], float), np.ascontiguousarray(anc, float)]
    lib().orc_eval_doppler(C.byref(f), _pp(params), T.dptr(r), _pp(J) if want_J else None)
    return r[0], J[:5]


def quat_plus(q, d):
    out = np.zeros(4)
    lib().orc_quat_plus(T.dptr(np.ascontiguousarray(q, float)), T.dptr(np.ascontiguousarray(d, float)), T.dptr(out))
    return out


def plane_qr_solve(A, b):
    x = np.zeros(3)
    lib().orc_plane_qr_solve(T.dptr(np.ascontiguousarray(A, float)), T.dptr(np.ascontiguousarray(b, float)), T.dptr(x))
    return x


def eval_binary_plane(cp, pnc, score, t1, q1, t2, q2):
    r = np.zeros(1)
    J = [np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(4)]
    cp = np.ascontiguousarray(cp, np.float32)
    lib().orc_eval_binary_plane(T.fptr(cp), T.dptr(np.ascontiguousarray(pnc, float)), C.c_double(score),
                                _pp([np.ascontiguousarray(a, float) for a in (t1, q1, t2, q2)]), T.dptr(r), _pp(J))
    return r[0], J


def batch_linearize(K, band, poses, ci, cj, cp, pnc, score):
    H = np.zeros((K, band + 1, 36))
    g = np.zeros((K, 6))
    cost = C.c_double()
    ok = lib().orc_batch_linearize(K, band, T.dptr(poses), C.c_int64(len(ci)), T.iptr(ci), T.iptr(cj), T.fptr(cp), T.dptr(pnc), T.dptr(score),
                                   T.dptr(H), T.dptr(g), C.byref(cost))
    assert ok
    return H, g, cost.value


# ------------------------------------------------------------------ the batch problem on keyframe poses
class OrcBatchProblem(C.Structure):
    _fields_ = [("K", C.c_int32), ("band", C.c_int32), ("n_con", C.c_int64), ("ci", T.c_int32_p), ("cj", T.c_int32_p), ("cp", T.c_float_p),
                ("norm_cent", T.c_double_p), ("score", T.c_double_p), ("n_dq", C.c_int32), ("dq_i", T.c_int32_p), ("dq_j", T.c_int32_p),
                ("dq_const", T.c_double_p), ("n_dd", C.c_int32), ("dd", C.POINTER(T.GlioDdPsr)), ("frame", T.GlioGnssFrame),
                ("n_imu", C.c_int32), ("pad_", C.c_int32), ("imu", C.POINTER(T.GlioPreint)), ("gravity", C.c_double),
                ("n_rp", C.c_int32), ("pad2_", C.c_int32), ("rp_i", T.c_int32_p), ("rp_j", T.c_int32_p), ("rp_const", T.c_double_p)]


def eval_delta_q(dq_const, qi, qj, want_J=True):
    r = np.zeros(3)
    J = [np.zeros((3, 4)), np.zeros((3, 4))]
    lib().orc_eval_delta_q(T.dptr(np.ascontiguousarray(dq_const, float)), _pp([np.ascontiguousarray(qi, float), np.ascontiguousarray(qj, float)]),
                           T.dptr(r), _pp(J) if want_J else None)
    return r, J


def eval_relative_pose(dq, dp, p1, q1, p2, q2, want_J=True):
    r = np.zeros(6)
    J = [np.zeros((6, 3)), np.zeros((6, 4)), np.zeros((6, 3)), np.zeros((6, 4))]
    lib().orc_eval_relative_pose(T.dptr(np.ascontiguousarray(dq, float)), T.dptr(np.ascontiguousarray(dp, float)),
                                 _pp([np.ascontiguousarray(a, float) for a in (p1, q1, p2, q2)]), T.dptr(r), _pp(J) if want_J else None)
    return r, J


class BatchProblem:
    """Owns the numpy buffers behind an orc_batch_problem: binary plane constraints, delta_q attitude constraints (i, j, const_diff),
    DD pseudorange factors (slot_i / slot_j = keyframe indices)."""

    def __init__(self, K, band, ci, cj, cp, nc, score, dq=None, dd=None, frame=None, imu=None, gravity=9.80511, rp=None):
        """imu: None (pose-only problem) or the K - 1 pre-integrations (dicts as synth.preintegrate returns, or GlioPreint) of the
        ImuFactor chain (Estimator.cpp:2990-3001); then the unknowns are 15 per keyframe."""
        self.K, self.band = K, band
        self.ci = np.ascontiguousarray(ci, np.int32); self.cj = np.ascontiguousarray(cj, np.int32)
        self.cp = np.ascontiguousarray(cp, np.float32); self.nc = np.ascontiguousarray(nc, np.float64); self.score = np.ascontiguousarray(score, np.float64)
        dq = dq or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 4)))
        self.dq_i = np.ascontiguousarray(dq[0], np.int32); self.dq_j = np.ascontiguousarray(dq[1], np.int32); self.dq_c = np.ascontiguousarray(dq[2], np.float64)
        dd = dd or []
        self.dd = (T.GlioDdPsr * max(len(dd), 1))(*dd)
        p = OrcBatchProblem()
        p.K, p.band, p.n_con = K, band, len(self.ci)
        p.ci, p.cj, p.cp, p.norm_cent, p.score = T.iptr(self.ci), T.iptr(self.cj), T.fptr(self.cp), T.dptr(self.nc), T.dptr(self.score)
        p.n_dq, p.dq_i, p.dq_j, p.dq_const = len(self.dq_i), T.iptr(self.dq_i), T.iptr(self.dq_j), T.dptr(self.dq_c)
        p.n_dd, p.dd = len(dd), self.dd
        if frame is not None:
            p.frame = frame
        imu = list(imu) if imu is not None else []
        assert len(imu) in (0, K - 1)
        self.imu = (T.GlioPreint * max(len(imu), 1))()
        for k, d in enumerate(imu):
            if isinstance(d, T.GlioPreint):
                self.imu[k] = d
            else:
                from glio_amd import synth
                synth.fill_preint(self.imu[k], d)
        p.n_imu, p.imu, p.gravity = len(imu), self.imu, gravity
        # rp = (i, j, const [n][7]): LidarPoseFactorBatchRelativeAutoDiff factors (batch.relative_pose_pairs)
        rp = rp or (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 7)))
        self.rp_i = np.ascontiguousarray(rp[0], np.int32); self.rp_j = np.ascontiguousarray(rp[1], np.int32); self.rp_c = np.ascontiguousarray(rp[2], np.float64)
        p.n_rp = len(self.rp_i)
        if p.n_rp:
            p.rp_i, p.rp_j, p.rp_const = T.iptr(self.rp_i), T.iptr(self.rp_j), T.dptr(self.rp_c)
        self.n_imu = len(imu)
        self.c = p

    @property
    def dim(self):
        return (15 if self.n_imu else 6) * self.K

    def linearize_dense(self, poses, speed_bias=None):
        """dense H, g, cost over [dt3 dtheta3 (dv3 dba3 dbg3)] per keyframe (orc_batch2_linearize)"""
        n = self.dim
        H = np.zeros((n, n)); g = np.zeros(n); cost = C.c_double()
        sb = np.ascontiguousarray(speed_bias, float) if self.n_imu else None
        ok = lib().orc_batch2_linearize(C.byref(self.c), T.dptr(np.ascontiguousarray(poses, float)), T.dptr(sb) if sb is not None else None, T.dptr(H), T.dptr(g), C.byref(cost))
        assert ok
        return H, g, cost.value

    def linearize_banded(self, poses, speed_bias=None):
        """the same without the dense matrix (K = 2000): lower band [n][hbw + 1] (entry (i, j), i - hbw <= j <= i, at [i][j - i + hbw]), g, cost"""
        n = self.dim
        lib().orc_batch2_half_bandwidth.restype = C.c_int
        hbw = lib().orc_batch2_half_bandwidth(C.byref(self.c))
        Hb = np.zeros((n, hbw + 1)); g = np.zeros(n); cost = C.c_double()
        sb = np.ascontiguousarray(speed_bias, float) if self.n_imu else None
        ok = lib().orc_batch2_linearize_banded(C.byref(self.c), T.dptr(np.ascontiguousarray(poses, float)), T.dptr(sb) if sb is not None else None, T.dptr(Hb), T.dptr(g), C.byref(cost))
        assert ok
        return Hb, g, cost.value

    def solve2(self, poses, opts, speed_bias=None, want_history=False):
        """orc_batch2_solve: returns (poses, speed_bias or None, summary[, history rows cost / radius / step norm / quality])"""
        x = np.ascontiguousarray(poses, float).copy()
        sb = np.ascontiguousarray(speed_bias, float).copy() if self.n_imu else None
        summ = T.GlioSummary()
        hist = np.zeros((opts.max_iterations + 1, 4))
        lib().orc_batch2_solve(C.byref(self.c), C.byref(opts), T.dptr(x), T.dptr(sb) if sb is not None else None, C.byref(summ), T.dptr(hist) if want_history else None)
        if want_history:
            return x, sb, summ, hist[:summ.iterations]
        return x, sb, summ

    def linearize(self, poses):
        K, band = self.K, self.band
        H = np.zeros((K, band + 1, 36)); g = np.zeros((K, 6)); cost = C.c_double()
        ok = lib().orc_batch_linearize_full(C.byref(self.c), T.dptr(np.ascontiguousarray(poses, float)), T.dptr(H), T.dptr(g), C.byref(cost))
        assert ok
        return H, g, cost.value

    def solve(self, poses, opts):
        x = np.ascontiguousarray(poses, float).copy()
        summ = T.GlioSummary()
        lib().orc_batch_solve(C.byref(self.c), C.byref(opts), T.dptr(x), C.byref(summ))
        return x, summ
