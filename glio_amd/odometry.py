"""Front-end scan-to-map odometry: host mirror of `LidarOdometry::updateTransformationWithCeres`
(reference GLIO/src/LidarOdometry.cpp:474-581) on the same C-ABI as the sliding window, with a one-keyframe window.

Per scan: kd-tree over `surf_from_map_ds` (:482) -> `match_cnt` rounds of [findCorrespondingSurfFeatures with the
current `abs_pose` (:343-404: gates 1.0 / 0.06 / 0.4), one Ceres problem of `LidarPlaneNormIncreFactor`s
(LidarKeyframeFactor.h:222-257: r = n^.(q p + t) + d^, no score, no extrinsic) under HuberLoss(0.1) and the Ceres
DEFAULT trust-region strategy (Levenberg-Marquardt), `max_num_iter` iterations (:505-530), quaternion sign
unification (:532-542)].  The reference also caps the solve at 15 ms wall time (:524) -- a non-deterministic
termination that is not restated; with max_num_iter = 12 (yaml:19) the iteration cap binds first on the GPU.
"""
import numpy as np

from . import ctypes_types as T
from . import synth


def frontend_opts(max_points, max_map_points, max_num_iter=12):
    """glio_opts of the front end: yaml `lidar_odometry` block + the constants of LidarOdometry.cpp."""
    o = synth.default_opts(1, pts=max(max_points, 64), map_pts=max(max_map_points, 64))
    o.max_iterations = max_num_iter                 # config_urban_hk.yaml:19
    o.kd_max_radius, o.surf_dist_thres, o.weight_gate = 1.0, 0.06, 0.4        # LidarOdometry.cpp:356,379,392
    o.huber_delta = 0.1                             # :499
    o.q_lb[:] = [1.0, 0, 0, 0]
    o.t_lb[:] = [0, 0, 0]                           # LidarPlaneNormIncreFactor applies no extrinsic
    o.unit_scores = 1
    o.trust_region_strategy = 1                     # Ceres default LEVENBERG_MARQUARDT (solverOptions :521-527)
    o.max_solver_time_s = 0.015                     # solverOptions.max_solver_time_in_seconds (:524)
    return o


LOCAL_MAP_WIDTH = 20          # `if (recent_surf_frames.size() < 20)`, LidarOdometry.cpp:278
LOCAL_MAP_LEAF = 0.2          # down_size_filter_surf_map.setLeafSize(0.2, 0.2, 0.2), :158


def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def _rotate(q, v):
    """Eigen's q * v (v + 2 w (u x v) + 2 u x (u x v)), in the operation order of glio::ScanToMapOdometry::rotate: the two hosts agree bit for bit"""
    uv = np.array([q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]])
    uv = uv + uv
    uuv = np.array([q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]])
    return np.array([v[k] + q[0] * uv[k] + uuv[k] for k in range(3)])


class ScanToMapOdometry:
    """`backend`: capi.Context (or a test double with the same methods) created with `frontend_opts`.  update() is updateTransformationWithCeres
    against a map the caller set; run() is LidarOdometry::run() (:661-699) per scan with the 20-frame / 0.2 m local map resident on the device --
    glio::ScanToMapOdometry (glio_backend.hpp) is the same class in C++ and documents the sequence."""

    def __init__(self, backend, scan_match_cnt=1):
        self.be = backend
        self.last = None
        self.scan_match_cnt = scan_match_cnt
        self.abs_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        self.rel_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        self.poses = 0
        self.map_points = None
        self._last_pose, self._last_cloud = None, None
        self._ring = False

    def _save(self, cloud):
        self._last_pose, self._last_cloud = self.abs_pose.copy(), np.ascontiguousarray(cloud, np.float32)
        self.poses += 1

    def run(self, surf_last_ds, max_points=None):
        """One scan (already downsampled: down_size_filter_surf, :312-313).  Returns (abs_pose, rounds)."""
        if not self._ring:
            self.be.localmap_config(LOCAL_MAP_WIDTH, LOCAL_MAP_LEAF, max_points or len(surf_last_ds))
            self._ring = True
        if self.poses == 0:                               # !system_initialized (:671-675)
            self._save(surf_last_ds)
            return self.abs_pose.copy(), []
        a, r = self.abs_pose, self.rel_pose               # poseInitialization (:405-432)
        t = _rotate(a[:4], r[4:]) + a[4:]
        self.abs_pose = np.r_[_qmul(a[:4], r[:4]), t]
        if self.poses <= 1:                               # buildLocalMap (:268-292) + downSampleCloud (:306-314)
            self.be.set_map(surf_last_ds); self.map_points = len(surf_last_ds)
        else:
            self.be.localmap_push(self._last_cloud, self._last_pose[:4], self._last_pose[4:])
            self.map_points = self.be.localmap_build()
        rounds = []
        if self.map_points >= 10:                         # (:477-480)
            self.abs_pose, rounds = self.update(surf_last_ds, self.abs_pose, match_cnt=8 if self.poses < 2 else self.scan_match_cnt)
        prev = self._last_pose
        self._save(surf_last_ds)
        n2 = float(prev[0] * prev[0] + prev[1] * prev[1] + prev[2] * prev[2] + prev[3] * prev[3])      # computeRelative (:434-471)
        qin = np.array([prev[0], -prev[1], -prev[2], -prev[3]]) / n2
        self.rel_pose = np.r_[_qmul(qin, self.abs_pose[:4]), _rotate(qin, self.abs_pose[4:] - prev[4:])]
        return self.abs_pose.copy(), rounds

    def set_map(self, surf_from_map_ds):
        self.be.set_map(surf_from_map_ds)           # kd_tree_surf_last->setInputCloud (:482)

    def update(self, surf_last_ds, abs_pose, match_cnt=1):
        """abs_pose = (q[4] w,x,y,z ; t[3]) as in the reference's abs_pose[7].  Returns the new abs_pose, and the
        (summary, kept count) of every matching round."""
        st = T.WindowState(1)
        st.quat[0] = np.asarray(abs_pose[:4], float)
        st.trans[0] = np.asarray(abs_pose[4:], float)
        st.n_ddt = 0
        self.be.set_imu([])
        self.be.set_prior(None)
        self.be.set_gnss(None, [], [])
        rounds = []
        self.be.set_scan(0, surf_last_ds)
        for _ in range(match_cnt):
            kept = self.be.associate_resident(0, st.quat[0], st.trans[0])
            st, summ = self.be.solve(st)
            if st.quat[0, 0] < 0:                   # unifyQuaternion (:532-542)
                st.quat[0] *= -1.0
            rounds.append((summ, kept))
        self.last = st
        return np.r_[st.quat[0], st.trans[0]], rounds
