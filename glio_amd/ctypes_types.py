"""ctypes mirrors of include/glio_types.h (plain-old-data buffer contracts of the hot path).

Field order and sizes must match the header exactly; tests/test_abi.py checks sizeof() of every
struct against the values the compiled library reports.
"""
import ctypes as C

import numpy as np

GLIO_DD_MAX_SAT = 20
BLK_TRANS, BLK_QUAT, BLK_SPEEDBIAS = 0, 1, 2

TERMINATION = {0: "NO_CONVERGENCE", 1: "FUNCTION_TOLERANCE", 2: "PARAMETER_TOLERANCE",
               3: "GRADIENT_TOLERANCE", 4: "MIN_RADIUS", 5: "FAILURE"}

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class GlioOpts(C.Structure):
    _fields_ = [
        ("window", C.c_int32), ("max_iterations", C.c_int32), ("max_points_per_scan", C.c_int32),
        ("max_map_points", C.c_int32), ("max_ddt_epochs", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("huber_delta", C.c_double), ("doppler_huber_delta", C.c_double),
        ("q_lb", C.c_double * 4), ("t_lb", C.c_double * 3),
        ("lidar_const", C.c_double), ("surf_dist_thres", C.c_double),
        ("kd_max_radius", C.c_double), ("weight_gate", C.c_double), ("gravity", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("trust_region_strategy", C.c_int32), ("unit_scores", C.c_int32),
        ("lidar_precision", C.c_int32), ("reserved_", C.c_int32), ("max_solver_time_s", C.c_double),
    ]


class GlioBatchTrOpts(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("use_nonmonotonic_steps", C.c_int32), ("max_consecutive_nonmonotonic_steps", C.c_int32),
                ("jacobi_scaling", C.c_int32), ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("dogleg_type", C.c_int32), ("reserved_", C.c_int32)]


DOGLEG_TRADITIONAL, DOGLEG_SUBSPACE = 0, 1


def batch_tr_opts(max_iterations=100, dogleg=DOGLEG_SUBSPACE):
    """ceres::Solver::Options of the batch solve (Estimator.cpp:3275-3281: DOGLEG, SUBSPACE_DOGLEG, non-monotonic steps; yaml
    max_num_iter: 100) + Ceres 1.14 defaults."""
    o = GlioBatchTrOpts()
    o.dogleg_type = dogleg
    o.max_iterations, o.use_nonmonotonic_steps, o.max_consecutive_nonmonotonic_steps, o.jacobi_scaling = max_iterations, 1, 5, 1
    o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius = 1e4, 1e16, 1e-32
    o.min_relative_decrease, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance = 1e-3, 1e-6, 1e-10, 1e-8
    return o


class GlioState(C.Structure):
    _fields_ = [("trans", c_double_p), ("quat", c_double_p), ("speed_bias", c_double_p),
                ("rcv_ddt", c_double_p), ("n_ddt", C.c_int32)]


class GlioPreint(C.Structure):
    _fields_ = [("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
                ("sum_dt", C.c_double), ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


class GlioPrior(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_blocks", C.c_int32), ("lin_jac", c_double_p), ("lin_res", c_double_p),
                ("blk_slot", c_int32_p), ("blk_kind", c_int32_p), ("blk_idx", c_int32_p), ("blk_x0", c_double_p)]


class GlioDdPsr(C.Structure):
    _fields_ = [("slot_i", C.c_int32), ("slot_j", C.c_int32), ("n_sat", C.c_int32), ("master", C.c_int32),
                ("ratio", C.c_double), ("threshold", C.c_double), ("station", C.c_double * 3),
                ("user_sat_pos", (C.c_double * 3) * GLIO_DD_MAX_SAT), ("ref_sat_pos", (C.c_double * 3) * GLIO_DD_MAX_SAT),
                ("user_psr", C.c_double * GLIO_DD_MAX_SAT), ("ref_psr", C.c_double * GLIO_DD_MAX_SAT),
                ("weight", C.c_double * ((GLIO_DD_MAX_SAT - 1) ** 2))]


class GlioDoppler(C.Structure):
    _fields_ = [("slot_i", C.c_int32), ("slot_j", C.c_int32), ("epoch", C.c_int32), ("pad_", C.c_int32),
                ("ratio", C.c_double), ("var", C.c_double), ("sat_pos", C.c_double * 3), ("sat_vel", C.c_double * 3),
                ("sv_ddt", C.c_double), ("doppler", C.c_double), ("lamda", C.c_double),
                ("lever_arm", C.c_double * 3), ("R_ecef_local", C.c_double * 9)]


class GlioGnssFrame(C.Structure):
    _fields_ = [("yaw_enu_local", C.c_double), ("anc_ecef", C.c_double * 3)]


class GlioSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("n_lidar_residuals", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double), ("gradient_max_norm", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["termination_name"] = TERMINATION.get(self.termination, "?")
        return d


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int32_p)


class WindowState:
    """tmpTrans / tmpQuat / tmpSpeedBias / para_rcv_ddt of the reference (Estimator.cpp:345-348,309)."""

    def __init__(self, W, n_ddt=0):
        self.W = W
        self.trans = np.zeros((W, 3))
        self.quat = np.zeros((W, 4))
        self.quat[:, 0] = 1.0
        self.speed_bias = np.zeros((W, 9))
        self.rcv_ddt = np.zeros(max(n_ddt, 1))
        self.n_ddt = n_ddt

    def copy(self):
        s = WindowState(self.W, self.n_ddt)
        s.trans[:] = self.trans
        s.quat[:] = self.quat
        s.speed_bias[:] = self.speed_bias
        s.rcv_ddt = self.rcv_ddt.copy()
        return s

    def c(self):
        return GlioState(dptr(self.trans), dptr(self.quat), dptr(self.speed_bias), dptr(self.rcv_ddt), self.n_ddt)


def preint_array(n):
    return (GlioPreint * max(n, 1))()
