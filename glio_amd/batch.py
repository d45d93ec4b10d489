"""Batch stage (scan-to-multiscan) driver: the only part of the path that shards over GPUs.

Host-side mirror of the relevant loop of `Estimator::optimizeBatchWithLandMark` (reference
GLIO/src/Estimator.cpp:2739-3410, constraints built at :3004-3076): every rank owns the constraints whose
source keyframe falls in its contiguous range, linearises them with the HIP kernel K8 into the block-banded
[H | g | cost] buffer, ONE all-reduce (RCCL over xGMI through torch.distributed on the device buffer) sums the
ranks, and every rank runs the same banded solve.  No collective anywhere else.
"""
import ctypes as C
import math

import numpy as np

from . import capi
from . import ctypes_types as T


def shard_range(K, rank, world, band=6):
    """Contiguous keyframe range [lo, hi) owned by `rank` (the source keyframe of a constraint decides): whole super-blocks of
    the block cyclic reduction (6 keyframes, 12 for bands > 6), glio_batch_shard_range restated (tests/test_batch_dist_cpu.py
    checks the two agree)."""
    sbk = 6 if band <= 6 else 12
    S = (K + sbk - 1) // sbk
    return min(K, (S * rank // world) * sbk), min(K, (S * (rank + 1) // world) * sbk)


def hg_size(K, band):
    return K * (band + 1) * 36 + K * 6 + 1


def unpack_hg(Hg, K, band):
    Hg = np.asarray(Hg)
    nH = K * (band + 1) * 36
    return Hg[:nH].reshape(K, band + 1, 36), Hg[nH:nH + 6 * K].reshape(K, 6), float(Hg[-1])


def dense_from_band(Hb, K, band):
    H = np.zeros((6 * K, 6 * K))
    for k in range(K):
        for d in range(band + 1):
            if k + d < K:
                blk = Hb[k, d].reshape(6, 6)
                H[6 * k:6 * k + 6, 6 * (k + d):6 * (k + d) + 6] = blk
                if d:
                    H[6 * (k + d):6 * (k + d) + 6, 6 * k:6 * k + 6] = blk.T
    return H


# ------------------------------------------------------------------ synthetic constraints (torch: CPU or GPU)
def _quat_to_R(q):
    import torch
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


def make_poses(K, seed=20260930, perturb=(0.05, 0.003)):
    """Ground-truth keyframe poses along a gently curving 1 m-spaced track and a perturbed initial guess."""
    rng = np.random.default_rng(seed)
    s = np.arange(K, dtype=np.float64)
    gt = np.zeros((K, 7))
    gt[:, 0] = s
    gt[:, 1] = 3.0 * np.sin(s / 40.0)
    gt[:, 2] = 1.5 + 0.2 * np.sin(s / 25.0)
    yaw = 0.075 * np.cos(s / 40.0)
    gt[:, 3] = np.cos(yaw / 2)
    gt[:, 6] = np.sin(yaw / 2)
    init = gt.copy()
    init[:, :3] += rng.normal(0, perturb[0], (K, 3))
    dth = rng.normal(0, perturb[1], (K, 3))
    for k in range(K):
        n = np.linalg.norm(dth[k])
        dq = np.r_[math.cos(n), math.sin(n) / n * dth[k]]
        w1, v1, w2, v2 = dq[0], dq[1:], gt[k, 3], gt[k, 4:]
        init[k, 3] = w1 * w2 - v1 @ v2
        init[k, 4:] = w1 * v2 + w2 * v1 + np.cross(v1, v2)
    return gt, init


BATCH_KF_DT = 0.125          # seconds between batch keyframes (1 m spacing at 8 m/s)


def make_batch_imu(K, seed=20260930, rate_per_kf=10, kf_dt=BATCH_KF_DT, perturb_v=0.05, noise=True):
    """The IMU chain of the batch problem (Estimator.cpp:2990-3001): one pre-integration per pair of consecutive keyframes
    of make_poses' track, from IMU samples of the analytic trajectory (position p(s), yaw(s), s = t / kf_dt), midpoint
    pre-integrated on the host exactly as the window generator does (synth.preintegrate = class Preintegration restated).
    Returns (preints [K - 1] dicts, speed_bias_gt [K][9], speed_bias_init [K][9]).  Edge k spans keyframes k .. k + 1
    (the consistent interval; the reference's own indexing looks off by one there)."""
    from . import synth
    rng = np.random.default_rng(seed + 31)

    def pos_d(s):       # first and second derivative of make_poses' position with respect to s
        d1 = np.array([1.0, 3.0 / 40.0 * math.cos(s / 40.0), 0.2 / 25.0 * math.cos(s / 25.0)])
        d2 = np.array([0.0, -3.0 / 1600.0 * math.sin(s / 40.0), -0.2 / 625.0 * math.sin(s / 25.0)])
        return d1, d2

    def Rz(s):
        y = 0.075 * math.cos(s / 40.0)
        c, sn = math.cos(y), math.sin(y)
        return np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1.0]])

    n = rate_per_kf
    dts = np.full(n, kf_dt / n)
    preints = []
    for k in range(K - 1):
        acc, gyr = np.zeros((n + 1, 3)), np.zeros((n + 1, 3))
        for i in range(n + 1):
            s = k + i / n
            d1, d2 = pos_d(s)
            a = d2 / (kf_dt * kf_dt)
            acc[i] = Rz(s).T @ (a + np.array([0, 0, synth.GRAVITY]))
            gyr[i] = [0.0, 0.0, -0.075 / 40.0 * math.sin(s / 40.0) / kf_dt]
        if noise:
            acc += rng.normal(0, synth.ACC_N, acc.shape); gyr += rng.normal(0, synth.GYR_N, gyr.shape)
        preints.append(synth.preintegrate(acc, gyr, dts, np.zeros(3), np.zeros(3)))
    sb_gt = np.zeros((K, 9))
    for k in range(K):
        sb_gt[k, :3] = pos_d(float(k))[0] / kf_dt
    sb_init = sb_gt.copy()
    sb_init[:, :3] += rng.normal(0, perturb_v, (K, 3))
    return preints, sb_gt, sb_init


def make_constraints(gt, lo, hi, per_kf, band, seed=20260930, device="cpu", search_range=None):
    """Pre-associated binary plane constraints of the source keyframes [lo, hi): `per_kf` per keyframe spread over
    its up-to 2*band neighbours, sorted by (ci, cj).  Returns host index arrays and torch data tensors on `device`.
    With `search_range` the neighbours are the reference's search windows instead (search_window: +-search_range in the interior, the
    2 search_range + 1 keyframes at either end of the batch for the first / last search_range keyframes, Estimator.cpp:3009-3017) --
    the structure whose END windows need band = 2 * search_range.
    Geometry follows the reference's construction (Estimator.cpp:3850-3857,3879-3884): point in frame ci,
    plane normal + 5-point centroid in the coordinates of frame cj, score = 2.5 * weight."""
    import torch
    K = len(gt)
    ci_list, cj_list = [], []
    for i in range(lo, hi):
        if search_range is None:
            nbr = [j for j in range(i - band, i + band + 1) if j != i and 0 <= j < K]
        else:
            s0 = search_window(i, K, search_range)
            nbr = [j for j in range(s0, s0 + 2 * search_range + 1) if j != i and 0 <= j < K and abs(j - i) <= band]
        share = [per_kf // len(nbr) + (1 if r < per_kf % len(nbr) else 0) for r in range(len(nbr))]
        for j, c in zip(nbr, share):
            ci_list.append(np.full(c, i, np.int32))
            cj_list.append(np.full(c, j, np.int32))
    ci = np.concatenate(ci_list) if ci_list else np.zeros(0, np.int32)
    cj = np.concatenate(cj_list) if cj_list else np.zeros(0, np.int32)
    n = len(ci)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919 * lo)
    gt_t = torch.as_tensor(gt, device=device)
    ti, tj = gt_t[torch.as_tensor(ci.astype(np.int64), device=device), :3], gt_t[torch.as_tensor(cj.astype(np.int64), device=device), :3]
    Ri = _quat_to_R(gt_t[torch.as_tensor(ci.astype(np.int64), device=device), 3:])
    Rj = _quat_to_R(gt_t[torch.as_tensor(cj.astype(np.int64), device=device), 3:])
    nw = torch.randn(n, 3, generator=g, device=device, dtype=torch.float64)
    nw = nw / nw.norm(dim=1, keepdim=True)
    cw = ti + (torch.rand(n, 3, generator=g, device=device, dtype=torch.float64) - 0.5) * 40.0
    off = torch.randn(n, 3, generator=g, device=device, dtype=torch.float64) * 2.0
    off = off - (off * nw).sum(1, keepdim=True) * nw
    pw = cw + off + nw * torch.randn(n, 1, generator=g, device=device, dtype=torch.float64) * 0.02
    p_i = torch.einsum("nji,nj->ni", Ri, pw - ti)                       # R_i^T (p_w - t_i)
    n_l = torch.einsum("nji,nj->ni", Rj, nw)
    c_l = torch.einsum("nji,nj->ni", Rj, cw - tj)
    cp = torch.zeros(n, 4, device=device, dtype=torch.float32)
    cp[:, :3] = p_i.to(torch.float32)
    nc = torch.cat([n_l, c_l], 1).contiguous()
    score = 2.5 * (0.5 + 0.5 * torch.rand(n, generator=g, device=device, dtype=torch.float64))
    return ci, cj, cp.contiguous(), nc, score.contiguous()


# ------------------------------------------------------------------ batch association
def search_window(idx, size, search_range, start_idx=0):
    """First keyframe of the 2*search_range+1 window searched for keyframe `idx` (Estimator.cpp:3009-3017):
    centred in the interior, clamped to the ends of the batch."""
    if idx >= search_range + start_idx and idx < size - 1 - search_range:
        return idx - search_range
    if idx < search_range + start_idx:
        return start_idx
    return size - 2 * search_range - 1


def pair_list(K, search_range):
    """All (idx, search_idx) pairs of a batch in (ci, cj) order: the loop of Estimator.cpp:3004-3076 /
    findGlobalCorrespondingSurfFeaturesAdd_Batch (:3814-3815)."""
    ci, cj = [], []
    for idx in range(K):
        s0 = search_window(idx, K, search_range)
        for j in range(s0, s0 + 2 * search_range + 1):
            if j != idx and 0 <= j < K:
                ci.append(idx); cj.append(j)
    return np.asarray(ci, np.int32), np.asarray(cj, np.int32)


def pair_shard(pair_ci, K, rank, world, band):
    """Slice [first, last) of a ci-major pair list (pair_list) owned by `rank`: the pairs whose SOURCE keyframe lies in the
    rank's keyframe range -- the same ownership rule as the constraints of the batch stage, so a rank associates exactly
    the pairs whose constraints it will linearise and nothing crosses ranks before the one all-reduce.  `band` is the STAGE's
    band (2 * search_range for pair_list(K, search_range): the end windows reach that far), because the ranges are cut on the
    super-blocks of the solver, whose size depends on it (shard_range)."""
    lo, hi = shard_range(K, rank, world, band)
    ci = np.asarray(pair_ci)
    return int(np.searchsorted(ci, lo, side="left")), int(np.searchsorted(ci, hi, side="left"))


def delta_q_pairs(odo, search_range, start_idx=0):
    """The attitude-constraint pairs of optimizeBatch (Estimator.cpp:2831-2891) from the odometry keyframe poses `odo` [K][7]
    (x y z qw qx qy qz): for every keyframe i, walk backward then forward adding (i, j, const_diff = q_i^-1 q_j) whenever the
    distance to the last taken keyframe exceeds `5 / search_range` (an INTEGER division in the reference: 0 for search_range > 5).
    The reference's quirks are kept: `factor_count` is reset only when it reaches search_range, so a backward walk that found
    fewer leaves its count (and its `p_tmp`) to the forward walk; q_i is sign-unified (w >= 0), q_j is not."""
    odo = np.asarray(odo, np.float64)
    K = len(odo)
    thr = float(5 // search_range)
    di, dj = [], []
    pos = odo[:, :3].tolist()          # the walk on plain floats, the quaternion products in one numpy batch afterwards (the per-step numpy
    for i in range(start_idx, K):      # calls made this function 0.4 s at K = 2000)
        px, py, pz = pos[i]
        count = 0
        for walk in (range(i, start_idx - 1, -1), range(i, K)):
            for j in walk:
                if count == search_range:
                    count = 0
                    break
                if j == i:
                    continue
                qx, qy, qz = pos[j]
                dx, dy, dz = px - qx, py - qy, pz - qz
                if (dx * dx + dy * dy + dz * dz) ** 0.5 > thr:
                    px, py, pz = qx, qy, qz
                    di.append(i); dj.append(j)
                    count += 1
    di = np.array(di, np.int32); dj = np.array(dj, np.int32)
    qi = odo[di, 3:] * np.where(odo[di, 3:4] < 0, -1.0, 1.0)             # q_i sign-unified (w >= 0), q_j is not
    qinv = np.concatenate([qi[:, :1], -qi[:, 1:]], axis=1) / np.sum(qi * qi, axis=1, keepdims=True)
    qj = odo[dj, 3:]
    w1, v1, w2, v2 = qinv[:, :1], qinv[:, 1:], qj[:, :1], qj[:, 1:]
    dc = np.concatenate([w1 * w2 - np.sum(v1 * v2, axis=1, keepdims=True), w1 * v2 + w2 * v1 + np.cross(v1, v2)], axis=1) if len(di) else np.zeros((0, 4))
    return di, dj, np.ascontiguousarray(dc, np.float64).reshape(-1, 4)


DDPSR_THRESHOLDS = (1e9, 10.0, 8.0, 6.0)      # Estimator.cpp:2764-2767: iteration_num = 4 rounds of the 7 listed values


def relative_pose_pairs(odo, search_range, start_idx=0):
    """The LidarPoseFactorBatchRelativeAutoDiff factors of optimizeBatch with sms_fusion_level == 0 (Estimator.cpp:2897-2955, the released default
    config_urban_hk.yaml:63) from the odometry keyframe poses `odo` [K][7] (x y z qw qx qy qz): for idx in [start + sr, K) and ms_i in 1..sr-1 the
    factor (idx - ms_i, idx), then for idx in [start, K - sr) and ms_i in 1..sr-1 the factor (idx, idx + ms_i) -- interior pairs appear in BOTH
    loops and are added twice, as in the reference.  delta_q = q_a^-1 q_b, delta_p = q_a^-1 (p_b - p_a) with Eigen's inverse() and q * v.
    Returns (i, j, const [n][7] = delta_q (w,x,y,z), delta_p)."""
    odo = np.asarray(odo, np.float64)
    K, sr = len(odo), search_range
    ri, rj = [], []
    for idx in range(start_idx + sr, K):
        for ms in range(1, sr):
            ri.append(idx - ms); rj.append(idx)
    for idx in range(start_idx, K - sr):
        for ms in range(1, sr):
            ri.append(idx); rj.append(idx + ms)
    ri = np.array(ri, np.int32); rj = np.array(rj, np.int32)
    if len(ri) == 0:
        return ri, rj, np.zeros((0, 7))
    qa, qb = odo[ri, 3:], odo[rj, 3:]
    qinv = np.concatenate([qa[:, :1], -qa[:, 1:]], axis=1) / np.sum(qa * qa, axis=1, keepdims=True)
    w1, v1, w2, v2 = qinv[:, :1], qinv[:, 1:], qb[:, :1], qb[:, 1:]
    dq = np.concatenate([w1 * w2 - np.sum(v1 * v2, axis=1, keepdims=True), w1 * v2 + w2 * v1 + np.cross(v1, v2)], axis=1)
    d = odo[rj, :3] - odo[ri, :3]
    uv = 2.0 * np.cross(v1, d)                                  # Eigen's _transformVector on the (non-normalised) inverse
    dp = d + w1 * uv + np.cross(v1, uv)
    return ri, rj, np.ascontiguousarray(np.concatenate([dq, dp], axis=1))


def make_batch_gnss(gt, seed=20260930, sats_per_sys=10, psr_sigma=1.0, outliers=0.05):
    """Synthetic double-differenced pseudorange factors of the batch problem: one GNSS epoch between every pair of consecutive
    keyframes (leftKey = k, rightKey = k + 1, ts_ratio as Estimator.cpp:3100-3130 derives it), two constellations, identity
    weight and the station position (addDDPsrResFactor_gl, Estimator.cpp:1899-1911); a fraction of the user pseudoranges carries
    a multipath-like outlier so that the {1e9, 10, 8, 6} threshold rounds have something to down-weight."""
    from . import synth
    rng = np.random.default_rng(seed + 9)
    K = len(gt)
    frame = T.GlioGnssFrame()
    frame.yaw_enu_local = 0.0
    frame.anc_ecef[:] = list(synth.ANCHOR_ECEF)
    Ree = synth.ecef2rotation(synth.ANCHOR_ECEF)
    up = synth.ANCHOR_ECEF / np.linalg.norm(synth.ANCHOR_ECEF)
    east, north = Ree[:, 0], Ree[:, 1]
    sats = []
    for _sys in range(2):
        pos = []
        while len(pos) < sats_per_sys:
            el, az = math.radians(rng.uniform(15, 85)), rng.uniform(0, 2 * math.pi)
            d = math.cos(el) * (math.sin(az) * east + math.cos(az) * north) + math.sin(el) * up
            b, c = synth.ANCHOR_ECEF @ d, synth.ANCHOR_ECEF @ synth.ANCHOR_ECEF - 26560e3 ** 2
            pos.append(synth.ANCHOR_ECEF + (-b + math.sqrt(b * b - c)) * d)
        sats.append(np.array(pos))
    dd = []
    for k in range(K - 1):
        ratio = float(rng.uniform(0.05, 0.95))
        Pe = Ree @ (ratio * gt[k, :3] + (1 - ratio) * gt[k + 1, :3]) + synth.ANCHOR_ECEF
        clock = 1234.5 + 0.3 * k
        for spos in sats:
            f = T.GlioDdPsr()
            f.slot_i, f.slot_j, f.n_sat = k, k + 1, sats_per_sys
            f.master = int(np.argmax([(sp - synth.ANCHOR_ECEF) @ up / np.linalg.norm(sp - synth.ANCHOR_ECEF) for sp in spos]))
            f.ratio, f.threshold = ratio, DDPSR_THRESHOLDS[0]
            f.station[:] = list(synth.STATION_ECEF)
            for i in range(sats_per_sys):
                f.user_sat_pos[i][:] = list(spos[i]); f.ref_sat_pos[i][:] = list(spos[i])
                bad = 25.0 * rng.uniform(0.5, 1.5) if (rng.uniform() < outliers and i != f.master) else 0.0
                f.user_psr[i] = np.linalg.norm(spos[i] - Pe) + clock + rng.normal(0, psr_sigma) + bad
                f.ref_psr[i] = np.linalg.norm(spos[i] - synth.STATION_ECEF) + 77.0 + rng.normal(0, 0.3)
            W = np.eye(sats_per_sys - 1)
            f.weight[:W.size] = list(W.ravel())
            dd.append(f)
    return dd, frame


def select_batch_gnss_epochs(obs_local_ts, keyframe_time, first_idx, n_poses, trans):
    """Which GNSS epochs enter the batch problem and between which keyframes (Estimator.cpp:3086-3126 with getGlobalLowerUpperIdx :1635-1663): rows
    (epoch, left_key, right_key, ts_ratio).  Same rules as glio::selectBatchGnssEpochs (glio_batch_backend.hpp), which documents them."""
    kt = np.asarray(keyframe_time, float)
    out, padd = [], np.zeros(3)
    # (pose index i reads kt[i - 1] and trans[i - 1]: first_idx = 0 would wrap to the LAST keyframe here and read out of bounds in the C++ twin)
    if first_idx < 1 or n_poses < first_idx or max(n_poses - 1, 0) > len(kt) or max(n_poses - 1, 0) > len(trans):
        raise ValueError("select_batch_gnss_epochs: need first_idx >= 1 and n_poses - 1 <= len(keyframe_time), len(trans)")
    if len(kt) == 0:
        return out
    for e, T in enumerate(np.asarray(obs_local_ts, float)):
        if T > kt[-1] or T < kt[0]:
            continue
        lower, upper, diff = -1, 10000000, 10000000.0
        for i in range(first_idx, n_poses):
            t = kt[i - 1]
            if abs(t - T) < diff and t < T:
                lower, diff = i, abs(t - T)
        diff = 10000000.0
        for i in range(first_idx, n_poses):
            t = kt[i - 1]
            if abs(t - T) < diff and t > T:
                upper, diff = i, abs(t - T)
        if not (0 <= lower < n_poses and 0 <= upper < n_poses):
            continue
        tl, tu = kt[lower - 1], kt[upper - 1]
        pj = np.asarray(trans[upper - 1], float)
        if np.linalg.norm(pj - padd) < 1.0:
            continue
        padd = pj.copy()
        out.append((e, lower - 1, upper - 1, (tu - T) / (tu - tl)))
    return out


def prn_system(prn):
    """gnss_tools.h:1116-1168: 0 GPS, 1 BeiDou, 2 GLONASS, 3 Galileo, -1 none"""
    if prn <= 32 or prn == 84:
        return 0
    if 87 <= prn <= 121:
        return 1
    if 32 < prn <= 56:
        return 2
    if 56 < prn < 87:
        return 3
    return -1


def dd_group(system, user_prn, user_psr, user_ele, ref_prn):
    """prepare<SYS>DDPsrData (Estimator.cpp:1702-1860): (rover indices, station indices, master) of one constellation; the master is chosen against a
    running maximum that is updated with the SIGNED elevation (quirk, replicated)."""
    user, ref = [], []
    for i, p in enumerate(user_prn):
        for j, q in enumerate(ref_prn):
            if p == q and prn_system(p) == system and user_psr[i] > 1000:
                user.append(i); ref.append(j)
    master, max_ele = -1, 0.0
    for m, i in enumerate(user):
        if abs(user_ele[i]) > max_ele:
            max_ele, master = user_ele[i], m
    return user, ref, master


def batch_selection_draws(count, res_num, rng, ends=False, rand_set_num=400):
    """The indices `globalFeatureSelectionAdd_Batch` (Estimator.cpp:4057-4116) keeps of a keyframe pair's `count` records:
    all of them (None) when count <= batch_feature_res_num (:4077), otherwise the first res_num entries of
    geneRandArrayNoRepeat(0, count - 1, res_num) -- a shuffle of the indices 0 .. count - 2, so the LAST record is never drawn
    (random_generator.hpp:79-93: n = high - low values) -- in shuffle order.
    ends = True: `globalFeatureSelection_Batch` (:3994-4055), used for the first / last search_range keyframes: nothing is
    selected when count - 1 < res_num or count < 50 -- and there the reference RETURNS, i.e. skips the remaining neighbours
    of that keyframe too (:4014-4017; the caller sees "stop" = the string 'return') --, the draw is the first res_num of a
    no-repeat random set of min(rand_set_num, count - 1, count - res_num - 1) indices (the same distribution)."""
    if not ends:
        if count <= res_num:
            return None
        return rng.choice(count - 1, size=res_num, replace=False).astype(np.int64)      # (the first res_num of a uniform shuffle, without shuffling 60 000 indices)
    if count - 1 < res_num or count < 50:
        return "return"
    rs = rand_set_num
    if count - 1 < rs:
        rs = count - 1
    if count - res_num < rs:
        rs = count - res_num - 1
    return rng.permutation(count - 1)[:min(res_num, max(rs, 0))].astype(np.int64)


class BatchAssociation:
    """Device-resident findGlobalCorrespondingSurfFeaturesAdd_Batch: keyframe clouds stay on the GPU, `run` builds the
    pair-major constraint arrays K8 consumes.  No CPU fallback."""

    def __init__(self, K, max_points_per_frame, max_constraints, device=0):
        lib = capi.load()
        if lib.glio_device_count() < 1:
            raise capi.GlioError("no HIP device visible: batch association has no CPU fallback")
        lib.glio_bassoc_destroy.restype = None
        self.K = K
        self._h = C.c_void_p()
        capi._check(lib.glio_bassoc_create(device, K, max_points_per_frame, C.c_int64(max_constraints), C.byref(self._h)))
        self.capacity = int(max_constraints)
        self.pair_ci = self.pair_cj = self.pair_count = None
        self.total = 0

    def close(self):
        if self._h:
            capi.load().glio_bassoc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_frame(self, k, scan):
        scan = np.ascontiguousarray(scan, np.float32)
        capi._check(capi.load().glio_bassoc_set_frame(self._h, k, T.fptr(scan) if len(scan) else None, len(scan)))

    def set_frame_strided(self, k, points, ioff):
        """surf_frames[k] from records of points.dtype.itemsize bytes (capi.PCL_XYZI: 32, intensity at 16)"""
        pts = np.ascontiguousarray(points)
        capi._check(capi.load().glio_bassoc_set_frame_strided(self._h, k, pts.ctypes.data_as(C.c_void_p) if len(pts) else None, len(pts), pts.dtype.itemsize, ioff))

    def run(self, poses, pair_ci, pair_cj):
        poses = np.ascontiguousarray(poses, np.float64)
        self.pair_ci = np.ascontiguousarray(pair_ci, np.int32); self.pair_cj = np.ascontiguousarray(pair_cj, np.int32)
        n = len(self.pair_ci)
        self.pair_count = np.zeros(max(n, 1), np.int64)
        tot = C.c_int64()
        capi._check(capi.load().glio_bassoc_run(self._h, T.dptr(poses), n, T.iptr(self.pair_ci) if n else None, T.iptr(self.pair_cj) if n else None,
                                                self.pair_count.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(tot)))
        self.pair_count = self.pair_count[:n]
        self.total = tot.value
        return self.pair_count, self.total

    # ---- batchFeatureAssociation (Estimator.cpp:3413-3432): the records of one keyframe's pairs ADDED behind what the object holds
    def reset(self):
        capi._check(capi.load().glio_bassoc_reset(self._h))
        self.total = 0

    def prepare(self, pair_ci, pair_cj):
        """Optional, ahead of a run whose pairs are known before its poses: build descriptors sent, hash tables of the search frames cleared."""
        ci = np.ascontiguousarray(pair_ci, np.int32); cj = np.ascontiguousarray(pair_cj, np.int32)
        capi._check(capi.load().glio_bassoc_prepare_async(self._h, len(ci), T.iptr(ci) if len(ci) else None, T.iptr(cj) if len(cj) else None))

    def set_frame_from_scan(self, k, ctx, slot, lidar_offset):
        """surf_frames[k] <- the scan resident in window slot `slot` of a capi.Context (device copy, minus the LiDAR offset)."""
        off = np.ascontiguousarray(lidar_offset, np.float32)
        capi._check(capi.load().glio_bassoc_set_frame_from_scan(self._h, k, ctx._h, slot, T.fptr(off)))

    def run_append(self, poses, pair_ci, pair_cj, wait=True):
        """Append the records of the given pairs; wait = False only enqueues (finish() returns the counts)."""
        poses = np.ascontiguousarray(poses, np.float64)
        ci = np.ascontiguousarray(pair_ci, np.int32); cj = np.ascontiguousarray(pair_cj, np.int32)
        n = len(ci)
        self._pending_n = n
        lib = capi.load()
        if not wait:
            capi._check(lib.glio_bassoc_run_append_async(self._h, T.dptr(poses), n, T.iptr(ci) if n else None, T.iptr(cj) if n else None))
            return None
        cnt = np.zeros(max(n, 1), np.int64); tot = C.c_int64()
        capi._check(lib.glio_bassoc_run_append(self._h, T.dptr(poses), n, T.iptr(ci) if n else None, T.iptr(cj) if n else None,
                                               cnt.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(tot)))
        self.total = tot.value
        return cnt[:n], self.total

    def select_tail_draws(self, res_num, raws):
        """globalFeatureSelectionAdd_Batch for the asynchronous run in flight, on its stream (glio_bassoc_select_tail_draws_async): raws = res_num uint64 per pair."""
        raws = np.ascontiguousarray(raws, np.uint64)
        capi._check(capi.load().glio_bassoc_select_tail_draws_async(self._h, int(res_num), raws.ctypes.data_as(C.POINTER(C.c_uint64))))

    def finish(self):
        n = getattr(self, "_pending_n", 0)
        cnt = np.zeros(max(n, 1), np.int64); tot = C.c_int64()
        capi._check(capi.load().glio_bassoc_finish(self._h, cnt.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(tot)))
        self.total = tot.value
        return cnt[:n], self.total

    def select_range(self, first, src, n_current):
        src = np.ascontiguousarray(src, np.int64)
        capi._check(capi.load().glio_bassoc_select_range(self._h, C.c_int64(first), C.c_int64(len(src)), src.ctypes.data_as(C.POINTER(C.c_int64)) if len(src) else None,
                                                         C.c_int64(n_current)))
        self.total = first + len(src)

    def select(self, res_num, rng, ends_of=None, rand_set_num=400):
        """Batch feature selection over the pairs of the last run (batch_feature_res_num records per pair at most), gathered on
        the device.  ends_of: optional predicate idx -> bool marking the source keyframes that take the `_Batch` (ends) rule."""
        offs = np.concatenate([[0], np.cumsum(self.pair_count)]).astype(np.int64)
        keep, counts = [], np.zeros(len(self.pair_count), np.int64)
        stopped = set()
        for p in range(len(self.pair_count)):
            c = int(self.pair_count[p])
            idx = int(self.pair_ci[p])
            ends = bool(ends_of(idx)) if ends_of is not None else False
            if ends and idx in stopped:
                d = None
            else:
                d = batch_selection_draws(c, res_num, rng, ends=ends, rand_set_num=rand_set_num)
                if isinstance(d, str):
                    stopped.add(idx); d = None
            sel = np.arange(c, dtype=np.int64) if d is None else d
            keep.append(offs[p] + sel); counts[p] = len(sel)
        src = np.ascontiguousarray(np.concatenate(keep) if keep else np.zeros(0, np.int64), np.int64)
        capi._check(capi.load().glio_bassoc_select(self._h, C.c_int64(len(src)), src.ctypes.data_as(C.POINTER(C.c_int64)) if len(src) else None, C.c_int64(self.total)))
        self.pair_count, self.total = counts, int(len(src))
        return src

    def read(self, first=0, n=None):
        n = self.total - first if n is None else n
        cp = np.zeros((max(n, 1), 4), np.float32); nc = np.zeros((max(n, 1), 6)); sc = np.zeros(max(n, 1))
        capi._check(capi.load().glio_bassoc_read(self._h, C.c_int64(first), C.c_int64(n), T.fptr(cp), T.dptr(nc), T.dptr(sc)))
        return cp[:n], nc[:n], sc[:n]

    def feed(self, stage):
        """Hand the device arrays to a BatchStage (K8) without leaving the GPU."""
        cp, nc, sc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib = capi.load()
        capi._check(lib.glio_bassoc_results_dev(self._h, C.byref(cp), C.byref(nc), C.byref(sc)))
        stage._keep = self
        capi._check(lib.glio_batch_set_constraints_pairs_dev(stage._h, len(self.pair_ci), T.iptr(self.pair_ci), T.iptr(self.pair_cj),
                                                             self.pair_count.ctypes.data_as(C.POINTER(C.c_int64)), cp, nc, sc))


class RoundsAssociation:
    """The association schedule of optimizeBatchWithLandMark's rounds (Estimator.cpp:3004-3076): the INTERIOR keyframes use the
    constraints stored when they left the sliding window (gl_vec_surf_*, filled by findGlobalCorrespondingSurfFeaturesAdd_Batch
    after each window solve -- here: associated once, at the poses given to `start`), the first / last `search_range` keyframes
    are re-searched in EVERY round at the current poses (findGlobalCorrespondingSurfFeatures_Batch, :3018-3030).  Three resident
    BatchAssociation objects (front ends, interior, back ends: the pair list in (ci, cj) order splits into these three runs);
    `__call__(poses)` re-runs the two end sets and hands the concatenated device arrays to the stage -- the `reassociate` hook of
    solve_batch_rounds.  (The random globalFeatureSelection_Batch draw is left to the caller, as everywhere.)"""

    def __init__(self, stage, scans, search_range, max_points_per_frame, device=0):
        self.stage, self.K, self.sr = stage, len(scans), search_range
        K = self.K
        ci, cj = pair_list(K, search_range)
        front = ci < search_range
        back = ci > K - 1 - search_range
        self.parts = []
        for mask in (front, ~(front | back), back):
            pci, pcj = ci[mask], cj[mask]
            ba = BatchAssociation(K, max_points_per_frame, max(1, int(mask.sum())) * max_points_per_frame, device=device)
            need = sorted(set(pci.tolist()) | set(pcj.tolist()))
            for k in need:
                ba.set_frame(k, scans[k])
            self.parts.append((ba, pci, pcj))
        self.device = device
        self.maxpts = max_points_per_frame
        self._buf = None
        self.runs = 0

    def close(self):
        for ba, _, _ in self.parts:
            ba.close()

    def _run(self, which, poses):
        ba, pci, pcj = self.parts[which]
        if len(pci):
            ba.run(poses, pci, pcj)
            self.runs += 1

    def _feed(self, changed=None):
        """Hand the three runs to the stage.  The records live in ONE set of device arrays -- the interior association's own result arrays when they have
        room behind the interior's records (the usual case: [interior | front region | back region], nothing of the interior is copied), else a separate
        set [front region | interior | back region] filled once at `start`; a round copies only the re-searched end runs into their fixed-capacity regions
        and tells the stage every pair's record range (glio_batch_update_constraints_pairs_at_dev) -- concatenating the three result sets anew every
        round cost 22 ms per round at K = 2000 (44 GB moved) against 2 ms for the re-search itself."""
        import torch
        dev = f"cuda:{self.device}"

        class _Dev:
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}

        lib = capi.load()

        def views(ba, n=None):
            cp, nc, sc = C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi._check(lib.glio_bassoc_results_dev(ba._h, C.byref(cp), C.byref(nc), C.byref(sc)))
            n = int(ba.total) if n is None else int(n)
            if n == 0:
                return None
            return (torch.as_tensor(_Dev(cp.value, (n, 4), "<f4"), device=dev), torch.as_tensor(_Dev(nc.value, (n, 6), "<f8"), device=dev),
                    torch.as_tensor(_Dev(sc.value, (n,), "<f8"), device=dev))
        first = changed is None or getattr(self, "_buf", None) is None
        if first:
            caps = [len(self.parts[0][1]) * self.maxpts, int(self.parts[1][0].total) if len(self.parts[1][1]) else 0, len(self.parts[2][1]) * self.maxpts]
            ntot = max(1, sum(caps))
            inter = self.parts[1][0]
            self._in_place = bool(len(self.parts[1][1])) and inter.capacity >= ntot
            if self._in_place:
                # The interior's records are tens of gigabytes at C4 size and they sit in the interior association's own result arrays, which have room
                # behind them (capacity = pairs x points, kept records are fewer): those arrays ARE the stage's arrays, laid out [interior | front region |
                # back region]; only the two small end runs are copied (20 ms per start for a 44 GB copy of the interior otherwise).
                self._base = [caps[1], 0, caps[1] + caps[0]]
                self._buf = views(inter, inter.capacity)
            else:
                self._base = [0, caps[0], caps[0] + caps[1]]
                # (an existing set of arrays is reused when it is large enough: a fresh 44 GB allocation costs 1.3 s of page-table set-up at C4 size,
                #  four times the association of all pairs itself -- scripts/batch_assoc_probe.py)
                if getattr(self, "_buf", None) is None or self._buf[0].shape[0] < ntot or getattr(self, "_buf_is_view", False):
                    self._buf = None
                    self.stage._keep = None
                    self._buf = (torch.empty((ntot, 4), dtype=torch.float32, device=dev), torch.empty((ntot, 6), dtype=torch.float64, device=dev),
                                 torch.empty((ntot,), dtype=torch.float64, device=dev))
            self._buf_is_view = self._in_place
        which_copy = (0, 1, 2) if first else tuple(changed)
        for w in which_copy:
            ba, pci, pcj = self.parts[w]
            if not len(pci) or (w == 1 and self._in_place):
                continue
            v = views(ba)
            if v is None:
                continue
            n, b0 = int(ba.total), self._base[w]
            for dst, src in zip(self._buf, v):
                dst[b0:b0 + n].copy_(src)
        cis, cjs, cnts, offs, chg = [], [], [], [], []
        for w, (ba, pci, pcj) in enumerate(self.parts):
            if not len(pci):
                continue
            cnt = np.asarray(ba.pair_count, np.int64)
            cis.append(np.asarray(pci, np.int32)); cjs.append(np.asarray(pcj, np.int32)); cnts.append(cnt)
            offs.append(self._base[w] + np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64))
            chg.append(np.full(len(pci), 1 if (changed is None or w in changed) else 0, np.uint8))
        # the three runs are consecutive in (ci, cj) order: the stage takes the PAIR list (one entry per keyframe pair), not a keyframe index per constraint
        pci = np.ascontiguousarray(np.concatenate(cis)); pcj = np.ascontiguousarray(np.concatenate(cjs)); pcount = np.ascontiguousarray(np.concatenate(cnts))
        poff = np.ascontiguousarray(np.concatenate(offs), np.int64)
        cp, nc, sc = self._buf
        _torch_done(cp)
        self.stage._keep = self._buf
        pchg = np.ascontiguousarray(np.concatenate(chg))
        # only the re-searched runs are marked as replaced: the stage keeps the moment records of the interior pairs (Estimator.cpp:3018-3030)
        capi._check(lib.glio_batch_update_constraints_pairs_at_dev(self.stage._h, len(pci), T.iptr(pci), T.iptr(pcj), pcount.ctypes.data_as(C.POINTER(C.c_int64)),
                                                                   poff.ctypes.data_as(C.POINTER(C.c_int64)),
                                                                   C.c_void_p(cp.data_ptr()), C.c_void_p(nc.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                                   pchg.ctypes.data_as(C.POINTER(C.c_uint8)) if changed is not None else None))
        self.n_constraints = int(pcount.sum())

    def start(self, poses):
        """all three sets at `poses` (the stored interior constraints are made here)"""
        for w in range(3):
            self._run(w, poses)
        self._feed()

    def __call__(self, poses):
        self._run(0, poses); self._run(2, poses)
        self._feed(changed=(0, 2))


# ------------------------------------------------------------------ the HIP stage
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)


def _torch_done(*tensors):
    """The library works on its OWN (non-blocking) streams, which are not ordered with torch's: a torch tensor handed to it must be complete first
    (torch.zeros / torch.cat / a generator's kernels may still be queued when the Python call returns)."""
    import torch
    for t in tensors:
        if hasattr(t, "is_cuda") and t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
            return


class BatchStage:
    def __init__(self, K, band, max_constraints, device=0):
        lib = capi.load()
        if lib.glio_device_count() < 1:
            raise capi.GlioError("no HIP device visible: the batch stage has no CPU fallback")
        lib.glio_batch_hg_size.restype = C.c_int64
        lib.glio_batch_destroy.restype = None
        self.K, self.band, self.device = K, band, device
        self._h = C.c_void_p()
        capi._check(lib.glio_batch_create(device, K, band, C.c_int64(max(1, max_constraints)), C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            capi.load().glio_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_constraints(self, ci, cj, cp, nc, score):
        """cp / nc / score: torch tensors on this GPU (borrowed, kept alive here) or numpy arrays (uploaded)."""
        ci = np.ascontiguousarray(ci, np.int32); cj = np.ascontiguousarray(cj, np.int32)
        n = len(ci)
        if hasattr(cp, "data_ptr"):
            assert cp.is_cuda and cp.dtype.is_floating_point and nc.is_contiguous() and cp.is_contiguous() and score.is_contiguous()
            self._keep = (cp, nc, score)
            _torch_done(cp)
            capi._check(capi.load().glio_batch_set_constraints_dev(self._h, C.c_int64(n), T.iptr(ci) if n else None, T.iptr(cj) if n else None,
                                                                    C.c_void_p(cp.data_ptr()), C.c_void_p(nc.data_ptr()), C.c_void_p(score.data_ptr())))
        else:
            cp = np.ascontiguousarray(cp, np.float32); nc = np.ascontiguousarray(nc, np.float64); score = np.ascontiguousarray(score, np.float64)
            capi._check(capi.load().glio_batch_set_constraints(self._h, C.c_int64(n), T.iptr(ci) if n else None, T.iptr(cj) if n else None,
                                                                T.fptr(cp) if n else None, T.dptr(nc) if n else None, T.dptr(score) if n else None))

    def set_constraints_pairs(self, pair_ci, pair_cj, pair_count, cp, nc, score, changed=None):
        """The constraint set as a PAIR list (pair p = keyframes (pair_ci[p], pair_cj[p]) with pair_count[p] consecutive records in the torch
        device tensors cp / nc / score, sorted by (ci, cj)).  `changed` (one flag per pair) marks the pairs whose records were replaced since the
        previous call; the moment records of the others are kept (glio_batch_update_constraints_pairs_dev)."""
        pci = np.ascontiguousarray(pair_ci, np.int32); pcj = np.ascontiguousarray(pair_cj, np.int32); cnt = np.ascontiguousarray(pair_count, np.int64)
        assert cp.is_cuda and nc.is_contiguous() and cp.is_contiguous() and score.is_contiguous()
        self._keep = (cp, nc, score)
        _torch_done(cp)
        chg = None if changed is None else np.ascontiguousarray(changed, np.uint8)
        capi._check(capi.load().glio_batch_update_constraints_pairs_dev(self._h, len(pci), T.iptr(pci), T.iptr(pcj), cnt.ctypes.data_as(C.POINTER(C.c_int64)),
                                                                        C.c_void_p(cp.data_ptr()), C.c_void_p(nc.data_ptr()), C.c_void_p(score.data_ptr()),
                                                                        chg.ctypes.data_as(C.POINTER(C.c_uint8)) if chg is not None else None))

    def new_hg(self):
        import torch
        out = torch.zeros(hg_size(self.K, self.band), dtype=torch.float64, device=f"cuda:{self.device}")
        _torch_done(out)          # (the fill must not land after the library's first write into the buffer)
        return out

    def linearize(self, poses, Hg):
        poses = np.ascontiguousarray(poses, np.float64)
        capi._check(capi.load().glio_batch_linearize_dev(self._h, T.dptr(poses), C.c_void_p(Hg.data_ptr())))

    def step(self, Hg, lam, poses):
        poses = np.ascontiguousarray(poses, np.float64)
        out = np.zeros_like(poses)
        md = C.c_double()
        capi._check(capi.load().glio_batch_step_dev(self._h, C.c_void_p(Hg.data_ptr()), C.c_double(lam), T.dptr(poses), T.dptr(out), C.byref(md)))
        return out, md.value

    def set_small_factors(self, dq=None, dd=None, frame=None, threshold=None, rp=None):
        """The replicated small factors every rank adds after the all-reduce: dq = (i, j, const_diff [n][4]) attitude constraints
        (delta_q_pairs), dd = list of GlioDdPsr between the bracketing keyframes, frame = GlioGnssFrame; `threshold` overrides
        every DD factor's DDpsrThreshold (the outer round's value); rp = (i, j, const [n][7]) relative-pose factors (relative_pose_pairs:
        the scan-to-multiscan constraints of sms_fusion_level 0)."""
        rp = rp if rp is not None else (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 7)))
        ri = np.ascontiguousarray(rp[0], np.int32); rj = np.ascontiguousarray(rp[1], np.int32); rc = np.ascontiguousarray(rp[2], np.float64)
        capi._check(capi.load().glio_batch_set_relative_pose_factors(self._h, len(ri), T.iptr(ri) if len(ri) else None, T.iptr(rj) if len(ri) else None,
                                                                     T.dptr(rc) if len(ri) else None))
        dq = dq if dq is not None else (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 4)))
        di = np.ascontiguousarray(dq[0], np.int32); dj = np.ascontiguousarray(dq[1], np.int32); dc = np.ascontiguousarray(dq[2], np.float64)
        dd = list(dd or [])
        if threshold is not None:
            for f in dd:
                f.threshold = float(threshold)
        arr = (T.GlioDdPsr * max(len(dd), 1))(*dd)
        capi._check(capi.load().glio_batch_set_small_factors(self._h, C.byref(frame) if frame is not None else None, len(di), T.iptr(di) if len(di) else None,
                                                             T.iptr(dj) if len(di) else None, T.dptr(dc) if len(di) else None, len(dd), arr if dd else None))

    def set_dd_threshold(self, threshold):
        capi._check(capi.load().glio_batch_set_dd_threshold(self._h, C.c_double(threshold)))

    def set_shard(self, rank, world):
        """This stage is rank `rank` of `world` (before set_small_factors); returns its keyframe range."""
        capi._check(capi.load().glio_batch_set_shard(self._h, rank, world))
        self.rank, self.world = rank, world
        return shard_range(self.K, rank, world, self.band)

    def set_imu(self, preints, gravity=None):
        """The ImuFactor chain (Estimator.cpp:2990-3001): K - 1 pre-integrations (dicts of synth.preintegrate or GlioPreint), [] removes it."""
        from . import synth
        preints = list(preints or [])
        arr = (T.GlioPreint * max(len(preints), 1))()
        for k, d in enumerate(preints):
            if isinstance(d, T.GlioPreint):
                arr[k] = d
            else:
                synth.fill_preint(arr[k], d)
        capi._check(capi.load().glio_batch_set_imu(self._h, len(preints), arr if preints else None, C.c_double(synth.GRAVITY if gravity is None else gravity)))
        self.n_imu = len(preints)

    def _hook(self, dist, on_allreduce=None):
        """The all-reduce hook of the library for torch.distributed: the collective is issued on the library's stream (made
        torch's current stream for the call), so it is ordered with the kernels around it and the host does not wait."""
        import torch
        if dist is None:
            return ALLREDUCE_FN(), None
        dev = f"cuda:{self.device}"
        calls = []

        class _Dev:            # a device buffer of the library seen through the CUDA array interface
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        # gloo (several processes sharing ONE GPU, where RCCL refuses duplicate devices: the hardware test of the multi-process
        # protocol on a one-GPU box) sums through a host copy -- stream-synchronous, which the contract allows
        staged = getattr(dist, "get_backend", None) is not None and getattr(dist, "is_initialized", lambda: False)() and dist.get_backend() == "gloo"

        views, streams = {}, {}        # the library passes the same few buffers and the same stream every iteration: wrap each once (the hook sits
                                       # on the host's enqueue path: ~5 calls per trust-region iteration)

        def hook(ptr, count, stream, user):
            t = views.get((ptr, count))
            if t is None:
                t = views[(ptr, count)] = torch.as_tensor(_Dev(ptr, count), device=dev)
            ext = streams.get(stream)
            if ext is None:
                ext = streams[stream] = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev)
            with torch.cuda.stream(ext):
                if staged:
                    h = t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    t.copy_(h)
                    ext.synchronize()
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                if on_allreduce:
                    on_allreduce(t)
            calls.append(int(count))

        return ALLREDUCE_FN(hook), calls

    def linearize_full(self, poses, speed_bias=None, dist=None):
        """diag(H), g, cost of the whole problem through the solver's path (shard + hook + assembly); n = 6 K or 15 K."""
        poses = np.ascontiguousarray(poses, np.float64)
        B = 15 if getattr(self, "n_imu", 0) else 6
        diag = np.zeros(B * self.K); g = np.zeros(B * self.K); cost = C.c_double()
        sb = np.ascontiguousarray(speed_bias, np.float64) if speed_bias is not None else None
        cb, _ = self._hook(dist)
        capi._check(capi.load().glio_batch_linearize_full(self._h, T.dptr(poses), T.dptr(sb) if sb is not None else None, cb, None, T.dptr(diag), T.dptr(g), C.byref(cost)))
        return diag, g, cost.value

    def add_small(self, poses, Hg):
        poses = np.ascontiguousarray(poses, np.float64)
        capi._check(capi.load().glio_batch_add_small_dev(self._h, T.dptr(poses), C.c_void_p(Hg.data_ptr())))

    def solve_tr(self, poses, opts=None, dist=None, on_allreduce=None, speed_bias=None):
        """ceres::Solve of the batch problem (Estimator.cpp:3275-3284) on the device, device resident.  With `dist`
        (torch.distributed or a stand-in with all_reduce, world > 1 set through set_shard) the library calls back five times per
        trust-region iteration with a small device buffer; the hook all-reduces it in place on the library's stream.
        Returns (poses, summary) or, with the IMU chain, (poses, speed_bias, summary)."""
        poses = np.ascontiguousarray(poses, np.float64).copy()
        opts = opts or T.batch_tr_opts()
        summ = T.GlioSummary()
        cb, calls = self._hook(dist, on_allreduce)
        sb = np.ascontiguousarray(speed_bias, np.float64).copy() if speed_bias is not None else None
        capi._check(capi.load().glio_batch_solve_tr2(self._h, T.dptr(poses), T.dptr(sb) if sb is not None else None, C.byref(opts), cb, None, C.byref(summ)))
        self.allreduces = len(calls) if calls is not None else 0
        self.allreduce_sizes = calls or []
        if sb is not None:
            return poses, sb, summ
        return poses, summ

    def counters(self):
        out = (C.c_int64 * 4)()
        capi._check(capi.load().glio_batch_debug_counters(self._h, out))
        return dict(hook_calls=out[0], hook_doubles=out[1], groups=out[2], bcr_levels=out[3])

    def time_solve(self, Hg, lam=1e-4, reps=5):
        ms = C.c_float()
        capi._check(capi.load().glio_batch_time_solve(self._h, C.c_void_p(Hg.data_ptr()), C.c_double(lam), reps, C.byref(ms)))
        return ms.value

    def set_solver(self, mode):
        """0 = sequential banded Cholesky (one workgroup), 1 = block cyclic reduction (default)."""
        capi._check(capi.load().glio_batch_debug_set_solver(self._h, mode))

    def linearize_mode(self, poses, Hg, mode):
        """test hook: K8 by streaming (0), by moments taken at `poses` (1), by the moments stored earlier evaluated at `poses` (2)"""
        poses = np.ascontiguousarray(poses, np.float64)
        capi._check(capi.load().glio_debug_batch_linearize_mode(self._h, T.dptr(poses), C.c_void_p(Hg.data_ptr()), int(mode)))

    def time_linearize_mode(self, poses, Hg, mode, reps=10):
        """mode 1: the pairs' moments taken at `poses` + their evaluation; mode 2: the evaluation alone (k_batch_moment_eval + assembly + cost)"""
        poses = np.ascontiguousarray(poses, np.float64)
        ms = C.c_float()
        capi._check(capi.load().glio_debug_batch_time_linearize_mode(self._h, T.dptr(poses), C.c_void_p(Hg.data_ptr()), int(mode), reps, C.byref(ms)))
        return ms.value

    def time_linearize(self, poses, Hg, reps=10):
        poses = np.ascontiguousarray(poses, np.float64)
        ms = C.c_float()
        capi._check(capi.load().glio_batch_time_linearize(self._h, T.dptr(poses), C.c_void_p(Hg.data_ptr()), reps, C.byref(ms)))
        return ms.value


class ShardedBatchSolve:
    """The driver of the sharded stage, the same object on one GPU, on 8 GPUs (RCCL) and in the CPU tests (gloo):
    `stage` is this rank's lineariser / solver -- anything with new_hg(), linearize(poses, Hg) and step(Hg, lam, poses), i.e.
    a BatchStage (HIP) or a stand-in -- holding only THIS rank's constraints; `dist` is torch.distributed (or None for one
    rank).  One all_reduce(SUM) of the [H band | g | cost] buffer per linearisation makes every rank hold the global system,
    then every rank takes the same step: no other collective."""

    def __init__(self, stage, dist=None, on_allreduce=None):
        self.stage, self.dist, self.on_allreduce = stage, dist, on_allreduce
        self._bufs = [stage.new_hg(), stage.new_hg()]
        self._flip = 0
        self.allreduces = 0

    def linearize(self, poses):
        self._flip ^= 1
        Hg = self._bufs[self._flip]
        self.stage.linearize(poses, Hg)
        if self.dist is not None:
            if self.on_allreduce is not None:
                self.on_allreduce(Hg)          # (bench.py times the collective here)
            else:
                self.dist.all_reduce(Hg, op=self.dist.ReduceOp.SUM)
            self.allreduces += 1
        return Hg, float(Hg[-1].item())

    def solve(self, poses0, iterations=10, lam=1e-4):
        return lm_solve(self.linearize, self.stage.step, poses0, iterations=iterations, lam=lam)


def lm_solve(linearize_reduced, step, poses0, iterations=10, lam=1e-4):
    """Damped Gauss-Newton loop of the batch stage.  `linearize_reduced(poses) -> (Hg, cost)` must return the
    ALL-REDUCED buffer (every rank sees the same numbers, so every rank takes the same decisions)."""
    poses = poses0.copy()
    Hg, cost = linearize_reduced(poses)
    history = [cost]
    for _ in range(iterations):
        cand, mdec = step(Hg, lam, poses)
        Hg2, cost2 = linearize_reduced(cand)
        if cost2 < cost:
            poses, Hg, cost = cand, Hg2, cost2
            lam = max(lam / 3.0, 1e-12)
        else:
            lam *= 4.0
        history.append(cost)
    return poses, history


def solve_batch_rounds(stage, poses0, odo, search_range, dd, frame, reassociate=None, opts=None, dist=None, thresholds=DDPSR_THRESHOLDS, speed_bias=None):
    """The outer loop of optimizeBatch (Estimator.cpp:2764-3410): `iteration_num` rounds, each re-searching the LiDAR
    correspondences at the current poses (`reassociate(poses)` must leave the new constraint set on `stage`; None keeps it),
    rebuilding the small factors with this round's DDpsr_threshold, and running one trust-region solve.  The attitude constraints
    are rebuilt every round from the ODOMETRY poses `odo` (pose_info_keyframe is not updated inside the loop).  With the IMU chain
    set on the stage, `speed_bias` [K][9] travels along and (poses, speed_bias, history) is returned."""
    import time
    poses = np.ascontiguousarray(poses0, np.float64).copy()
    sb = None if speed_bias is None else np.ascontiguousarray(speed_bias, np.float64).copy()
    dq = delta_q_pairs(odo, search_range)
    stage.set_small_factors(dq, dd, frame, threshold=thresholds[0])
    history = []
    for thr in thresholds:
        if reassociate is not None:
            reassociate(poses)
        stage.set_dd_threshold(thr)
        t0 = time.perf_counter()
        if sb is not None:
            poses, sb, summ = stage.solve_tr(poses, opts, dist, speed_bias=sb)
        else:
            poses, summ = stage.solve_tr(poses, opts, dist)
        h = summ.as_dict()
        h["solve_ms"] = (time.perf_counter() - t0) * 1e3
        history.append(h)
    if sb is not None:
        return poses, sb, history
    return poses, history


class ThreadRanks:
    """N "virtual ranks" inside one process, one thread each, for exercising the sharded solve where only one GPU (or none: the
    CPU stand-in) is available: all_reduce = device synchronise, barrier, rank 0 sums the buffers in rank order and writes the sum
    back to every rank, barrier.  Same call sequence as torch.distributed; used by tests/ and by bench.py's projection."""

    class ReduceOp:
        SUM = "sum"

    def __init__(self, world, sync=None):
        import threading
        self.world, self.sync = world, sync
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = [0] * world
        self.errors = []

    def view(self, rank):
        outer = self

        class _Rank:
            ReduceOp = outer.ReduceOp

            def all_reduce(self, t, op=None):
                if outer.sync:
                    outer.sync()
                outer.slots[rank] = t
                outer.barrier.wait()
                if rank == 0:
                    tot = outer.slots[0].clone()
                    for q in outer.slots[1:]:
                        tot += q
                    for q in outer.slots:
                        q.copy_(tot)
                    if outer.sync:
                        outer.sync()
                outer.barrier.wait()
                outer.calls[rank] += 1

        return _Rank()

    def run(self, fn):
        """fn(rank, dist_view) in `world` threads; returns the list of results (exceptions are re-raised)."""
        import threading
        out = [None] * self.world

        def work(r):
            try:
                out[r] = fn(r, self.view(r))
            except BaseException as e:      # noqa: BLE001 -- re-raised below
                self.errors.append(e)
                self.barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if self.errors:
            raise self.errors[0]
        return out
