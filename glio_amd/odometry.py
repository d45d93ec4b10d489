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


class ScanToMapOdometry:
    """`backend`: capi.Context (or a test double with the same methods) created with `frontend_opts`."""

    def __init__(self, backend):
        self.be = backend
        self.last = None

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
