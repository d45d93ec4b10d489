"""Independent numpy transcription of the reference's IMU / marginalization formulas, written in the
reference's own matrix vocabulary (Qleft/Qright/LeftQuatMatrix of GLIO/include/utils/math_tools.h)
rather than the expanded scalar form the C oracle uses.  Two transcriptions agreeing is the pin we
have in place of reference golden vectors (there are none -- SURVEY.md section 4)."""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def Qleft(q):      # math_tools.h:36-42
    M = np.zeros((4, 4))
    M[0, 0] = q[0]; M[0, 1:] = -q[1:]; M[1:, 0] = q[1:]; M[1:, 1:] = q[0] * np.eye(3) + skew(q[1:])
    return M


def Qright(p):     # math_tools.h:45-51
    M = np.zeros((4, 4))
    M[0, 0] = p[0]; M[0, 1:] = -p[1:]; M[1:, 0] = p[1:]; M[1:, 1:] = p[0] * np.eye(3) - skew(p[1:])
    return M


def LeftQuatMatrix(q):   # math_tools.h:141-150 (vector-first layout)
    M = np.zeros((4, 4))
    M[:3, :3] = q[0] * np.eye(3) + skew(q[1:]); M[3, :3] = -q[1:]; M[:3, 3] = q[1:]; M[3, 3] = q[0]
    return M


def qmul(a, b):
    return Qleft(a) @ b


def qinv(q):
    return np.r_[q[0], -q[1:]] / (q @ q)


def rot(q, v):     # Eigen _transformVector
    uv = 2 * np.cross(q[1:], v)
    return v + q[0] * uv + np.cross(q[1:], uv)


def toR(q):
    return np.column_stack([rot(q, e) for e in np.eye(3)]) if abs(q @ q - 1) < 1e-12 else None


def imu_factor(pre, params, gravity):
    """ImuFactor::Evaluate (ImuFactor.h:21-171) + Preintegration::evaluate (Preintegration.h:196-235)."""
    O_P, O_R, O_V, O_BA, O_BG = 0, 3, 6, 9, 12
    Pi, Qi, SBi, Pj, Qj, SBj = [np.asarray(p, float) for p in params]
    Qi = Qi / np.linalg.norm(Qi); Qj = Qj / np.linalg.norm(Qj)
    Vi, Bai, Bgi = SBi[:3], SBi[3:6], SBi[6:]
    Vj, Baj, Bgj = SBj[:3], SBj[3:6], SBj[6:]
    Jm = np.asarray(pre["jacobian"]).reshape(15, 15)
    dp_dba, dp_dbg = Jm[O_P:O_P + 3, O_BA:O_BA + 3], Jm[O_P:O_P + 3, O_BG:O_BG + 3]
    dq_dbg = Jm[O_R:O_R + 3, O_BG:O_BG + 3]
    dv_dba, dv_dbg = Jm[O_V:O_V + 3, O_BA:O_BA + 3], Jm[O_V:O_V + 3, O_BG:O_BG + 3]
    g = np.array([0, 0, -gravity]); dt = pre["sum_dt"]
    dba, dbg = Bai - pre["linearized_ba"], Bgi - pre["linearized_bg"]
    cdq = qmul(np.asarray(pre["delta_q"]), np.r_[1.0, dq_dbg @ dbg / 2])
    cdv = pre["delta_v"] + dv_dba @ dba + dv_dbg @ dbg
    cdp = pre["delta_p"] + dp_dba @ dba + dp_dbg @ dbg
    Qi_inv = qinv(Qi)
    r = np.zeros(15)
    tmp = -0.5 * g * dt * dt + Pj - Pi - Vi * dt
    tmp1 = -g * dt + Vj - Vi
    r[O_P:O_P + 3] = rot(Qi_inv, tmp) - cdp
    qe = qmul(qinv(cdq), qmul(Qi_inv, Qj)); qe = qe / np.linalg.norm(qe)
    r[O_R:O_R + 3] = 2 * qe[1:]
    r[O_V:O_V + 3] = rot(Qi_inv, tmp1) - cdv
    r[O_BA:O_BA + 3] = Baj - Bai
    r[O_BG:O_BG + 3] = Bgj - Bgi
    cov = np.asarray(pre["covariance"]).reshape(15, 15)
    S = np.linalg.cholesky(np.linalg.inv(cov)).T
    Ri_inv = np.column_stack([rot(Qi_inv, e) for e in np.eye(3)])
    w, u = Qi[0], Qi[1:]
    J = [np.zeros((15, s)) for s in (3, 4, 9, 3, 4, 9)]
    J[0][O_P:O_P + 3] = -Ri_inv
    for row, v in ((O_P, tmp), (O_V, tmp1)):
        J[1][row:row + 3, 0] = 2 * (w * v + skew(u) @ v)
        J[1][row:row + 3, 1:] = 2 * ((u @ v) * np.eye(3) + np.outer(u, v) - np.outer(v, u) - w * skew(v))
    J[1][O_R:O_R + 3] = -2 * (Qleft(qinv(Qj)) @ Qright(cdq))[1:, :]
    J[2][O_P:O_P + 3, 0:3] = -Ri_inv * dt; J[2][O_P:O_P + 3, 3:6] = -dp_dba; J[2][O_P:O_P + 3, 6:9] = -dp_dbg
    J[2][O_R:O_R + 3, 6:9] = -LeftQuatMatrix(qmul(qmul(qinv(Qj), Qi), cdq))[:3, :3] @ dq_dbg
    J[2][O_V:O_V + 3, 0:3] = -Ri_inv; J[2][O_V:O_V + 3, 3:6] = -dv_dba; J[2][O_V:O_V + 3, 6:9] = -dv_dbg
    J[2][O_BA:O_BA + 3, 3:6] = -np.eye(3); J[2][O_BG:O_BG + 3, 6:9] = -np.eye(3)
    J[3][O_P:O_P + 3] = Ri_inv
    J[4][O_R:O_R + 3] = 2 * Qleft(qmul(qinv(cdq), Qi_inv))[1:, :]
    J[5][O_V:O_V + 3, 0:3] = Ri_inv; J[5][O_BA:O_BA + 3, 3:6] = np.eye(3); J[5][O_BG:O_BG + 3, 6:9] = np.eye(3)
    return S @ r, [S @ j for j in J]


def quat_plus(q, d):
    n = np.linalg.norm(d)
    dq = np.r_[np.cos(n), np.sin(n) / n * d] if n > 0 else np.array([1.0, 0, 0, 0])
    return qmul(dq, q)


def plus_jacobian(q):
    """d([1,delta] (x) q)/d delta -- Ceres QuaternionParameterization::ComputeJacobian."""
    return np.vstack([-q[1:], q[0] * np.eye(3) - skew(q[1:])])
