/*
 * orc_batch.c -- CPU ORACLE (test infrastructure): one linearisation of the batch stage's
 * scan-to-multiscan constraints (Estimator::optimizeBatchWithLandMark, GLIO/src/Estimator.cpp:
 * 3004-3076) -- BinaryLidarPlaneNormFactor (LidarKeyframeFactor.h:124-164), no loss function
 * (:2768), Ceres QuaternionParameterization on both quaternion blocks -- summed into the block-banded
 * normal equations that the multi-GPU path all-reduces.  Factor evaluators pinned on the reference's own code (oracle/_ref, tests/test_oracle_ref.py); see glio_oracle.h.
 */
#include <stdlib.h>
#include "glio_oracle.h"
#include "orc_math.h"

static void plusJ(const double q[4], double P[12]) {
    P[0] = -q[1]; P[1] = -q[2]; P[2] = -q[3];
    P[3] = q[0];  P[4] = q[3];  P[5] = -q[2];
    P[6] = -q[3]; P[7] = q[0];  P[8] = q[1];
    P[9] = q[2];  P[10] = -q[1]; P[11] = q[0];
}

int orc_batch_linearize(int K, int band, const double* poses, int64_t n_con, const int32_t* ci,
                        const int32_t* cj, const float* cp, const double* norm_cent,
                        const double* score, double* Hband, double* g, double* cost_out) {
    const size_t hb = (size_t)K * (band + 1) * 36;
    memset(Hband, 0, sizeof(double) * hb);
    memset(g, 0, sizeof(double) * (size_t)K * 6);
    double cost = 0;
    for (int64_t c = 0; c < n_con; ++c) {
        const int a = ci[c], b = cj[c];
        if (a == b || abs(a - b) > band) return 0;
        const double* P[4] = {poses + 7 * (size_t)a, poses + 7 * (size_t)a + 3, poses + 7 * (size_t)b, poses + 7 * (size_t)b + 3};
        double r, J0[3], J1[4], J2[3], J3[4];
        double* J[4] = {J0, J1, J2, J3};
        orc_eval_binary_plane(cp + 4 * (size_t)c, norm_cent + 6 * (size_t)c, score[c], P, &r, J);
        cost += 0.5 * r * r;
        double Pa[12], Pb[12], Ja[6], Jb[6];
        plusJ(P[1], Pa); plusJ(P[3], Pb);
        for (int k = 0; k < 3; ++k) {
            Ja[k] = J0[k]; Jb[k] = J2[k];
            Ja[3 + k] = J1[0] * Pa[k] + J1[1] * Pa[3 + k] + J1[2] * Pa[6 + k] + J1[3] * Pa[9 + k];
            Jb[3 + k] = J3[0] * Pb[k] + J3[1] * Pb[3 + k] + J3[2] * Pb[6 + k] + J3[3] * Pb[9 + k];
        }
        double* Haa = Hband + ((size_t)a * (band + 1) + 0) * 36;
        double* Hbb = Hband + ((size_t)b * (band + 1) + 0) * 36;
        const int lo = a < b ? a : b, d = abs(a - b);
        double* Hlo = Hband + ((size_t)lo * (band + 1) + d) * 36;   /* block (lo, lo+d) */
        const double* Jlo = a < b ? Ja : Jb;
        const double* Jhi = a < b ? Jb : Ja;
        for (int u = 0; u < 6; ++u) {
            g[6 * (size_t)a + u] += Ja[u] * r;
            g[6 * (size_t)b + u] += Jb[u] * r;
            for (int v = 0; v < 6; ++v) {
                Haa[u * 6 + v] += Ja[u] * Ja[v];
                Hbb[u * 6 + v] += Jb[u] * Jb[v];
                Hlo[u * 6 + v] += Jlo[u] * Jhi[v];
            }
        }
    }
    *cost_out = cost;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * delta_q_factor_auto (LidarKeyframeFactor.h:283-303):
 *   residual = 10000 (dq^-1 * qi^-1 * qj).vec          (Eigen inverse = conjugate / squaredNorm)
 * p = A (x) u (x) v with A = dq^-1, u = qi^-1, v = qj:  dp/dv = Qleft(A (x) u),  dp/du = Qleft(A) Qright(v),
 * du/dqi = (C - 2 (C qi) qi^T / |qi|^2) / |qi|^2 with C = diag(1,-1,-1,-1).
 */
int orc_eval_delta_q(const double dq_const[4], double const* const* P, double* r, double** J) {
    const double* qi = P[0];
    const double* qj = P[1];
    double A[4], u[4], Au[4], p[4];
    q_inv(dq_const, A);
    q_inv(qi, u);
    q_mul(A, u, Au);
    q_mul(Au, qj, p);
    for (int k = 0; k < 3; ++k) r[k] = 10000.0 * p[1 + k];
    if (!J) return 1;
    if (J[1]) {
        double L[16];
        q_left(Au, L);
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) J[1][k * 4 + c] = 10000.0 * L[(1 + k) * 4 + c];
    }
    if (J[0]) {
        double LA[16], Rv[16], M[16], dU[16];
        q_left(A, LA); q_right(qj, Rv);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { double s = 0; for (int k = 0; k < 4; ++k) s += LA[a * 4 + k] * Rv[k * 4 + b]; M[a * 4 + b] = s; }
        const double n2 = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
        const double Cq[4] = {qi[0], -qi[1], -qi[2], -qi[3]};
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) dU[a * 4 + b] = ((a == b ? (a == 0 ? 1.0 : -1.0) : 0.0) - 2.0 * Cq[a] * qi[b] / n2) / n2;
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) { double s = 0; for (int m = 0; m < 4; ++m) s += M[(1 + k) * 4 + m] * dU[m * 4 + c]; J[0][k * 4 + c] = 10000.0 * s; }
    }
    return 1;
}

/* LidarPoseFactorBatchRelativeAutoDiff (LidarPoseFactor.h:55-97; built at Estimator.cpp:2897-2955 for sms_fusion_level == 0, the shipped
 * default config_urban_hk.yaml:63): blocks P1[3], Q1[4], P2[3], Q2[4]; 6 residuals
 *   r[0:3] = 10 * 2 (dq^-1 * Q1^-1 * Q2).vec,   r[3:6] = 20 * (Q1^-1 * (P2 - P1) - dp)
 * with Eigen's inverse() (conjugate / squared norm) and Eigen's q * v = v + 2 w (u x v) + 2 u x (u x v) on the NON-normalised Q1^-1 --
 * exactly what the Jets differentiate.  Global Jacobians 6 x {3, 4, 3, 4}, row-major. */
int orc_eval_relative_pose(const double dq[4], const double dp[3], double const* const* P, double* r, double** J) {
    const double *p1 = P[0], *q1 = P[1], *p2 = P[2], *q2 = P[3];
    double A[4], u[4], Au[4], p[4], v[3], rv[3];
    q_inv(dq, A);
    q_inv(q1, u);
    q_mul(A, u, Au);
    q_mul(Au, q2, p);
    for (int k = 0; k < 3; ++k) { v[k] = p2[k] - p1[k]; r[k] = 10.0 * 2.0 * p[1 + k]; }
    q_rot(u, v, rv);
    for (int k = 0; k < 3; ++k) r[3 + k] = 20.0 * (rv[k] - dp[k]);
    if (!J) return 1;
    /* M(u) = d (u * v) / d v = I + 2 w [q]x + 2 [q]x [q]x */
    double Sq[9], Sq2[9], M[9];
    skew3(u + 1, Sq);
    mat_mul(Sq, Sq, Sq2, 3, 3, 3);
    for (int k = 0; k < 9; ++k) M[k] = ((k % 4 == 0) ? 1.0 : 0.0) + 2.0 * u[0] * Sq[k] + 2.0 * Sq2[k];
    if (J[0]) { for (int k = 0; k < 9; ++k) J[0][k] = 0.0; for (int k = 0; k < 9; ++k) J[0][9 + k] = -20.0 * M[k]; }
    if (J[2]) { for (int k = 0; k < 9; ++k) J[2][k] = 0.0; for (int k = 0; k < 9; ++k) J[2][9 + k] = 20.0 * M[k]; }
    if (J[3]) {
        double L[16];
        q_left(Au, L);
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) { J[3][k * 4 + c] = 20.0 * L[(1 + k) * 4 + c]; J[3][(3 + k) * 4 + c] = 0.0; }
    }
    if (J[1]) {
        /* d u / d q1, u = conj(q1) / |q1|^2 */
        double dU[16];
        const double n2 = q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2] + q1[3] * q1[3];
        const double Cq[4] = {q1[0], -q1[1], -q1[2], -q1[3]};
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) dU[a * 4 + b] = ((a == b ? (a == 0 ? 1.0 : -1.0) : 0.0) - 2.0 * Cq[a] * q1[b] / n2) / n2;
        /* rows 0..2: 20 [L(A) R(q2)]_(vec rows) dU */
        double LA[16], Rv[16], Mq[16];
        q_left(A, LA); q_right(q2, Rv);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { double s = 0; for (int k = 0; k < 4; ++k) s += LA[a * 4 + k] * Rv[k * 4 + b]; Mq[a * 4 + b] = s; }
        /* rows 3..5: 20 d(u * v)/d u dU;  d/dw = 2 (q x v),  d/dq = -2 w [v]x + 2 ((q.v) I + q v^T - 2 v q^T) */
        double D[12], qxv[3], Sv[9];
        v3_cross(u + 1, v, qxv);
        skew3(v, Sv);
        const double qv = v3_dot(u + 1, v);
        for (int k = 0; k < 3; ++k) {
            D[k * 4 + 0] = 2.0 * qxv[k];
            for (int c = 0; c < 3; ++c)
                D[k * 4 + 1 + c] = -2.0 * u[0] * Sv[k * 3 + c] + 2.0 * ((k == c ? qv : 0.0) + u[1 + k] * v[c] - 2.0 * v[k] * u[1 + c]);
        }
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) {
            double s0 = 0, s1 = 0;
            for (int m = 0; m < 4; ++m) { s0 += Mq[(1 + k) * 4 + m] * dU[m * 4 + c]; s1 += D[k * 4 + m] * dU[m * 4 + c]; }
            J[1][k * 4 + c] = 20.0 * s0; J[1][(3 + k) * 4 + c] = 20.0 * s1;
        }
    }
    return 1;
}

/* adds J^T J / J^T r of one residual block with local Jacobians Ja (nr x 6, keyframe a) and Jb (nr x 6, keyframe b) */
static void band_add(int band, double* Hband, double* g, int a, int b, int nr, const double* Ja, const double* Jb, const double* r) {
    double* Haa = Hband + ((size_t)a * (band + 1) + 0) * 36;
    double* Hbb = Hband + ((size_t)b * (band + 1) + 0) * 36;
    const int lo = a < b ? a : b, d = abs(a - b);
    double* Hlo = Hband + ((size_t)lo * (band + 1) + d) * 36;
    const double* Jlo = a < b ? Ja : Jb;
    const double* Jhi = a < b ? Jb : Ja;
    for (int u = 0; u < 6; ++u) {
        double ga = 0, gb = 0;
        for (int i = 0; i < nr; ++i) { ga += Ja[i * 6 + u] * r[i]; gb += Jb[i * 6 + u] * r[i]; }
        g[6 * (size_t)a + u] += ga; g[6 * (size_t)b + u] += gb;
        for (int v = 0; v < 6; ++v) {
            double saa = 0, sbb = 0, slh = 0;
            for (int i = 0; i < nr; ++i) { saa += Ja[i * 6 + u] * Ja[i * 6 + v]; sbb += Jb[i * 6 + u] * Jb[i * 6 + v]; slh += Jlo[i * 6 + u] * Jhi[i * 6 + v]; }
            Haa[u * 6 + v] += saa; Hbb[u * 6 + v] += sbb; Hlo[u * 6 + v] += slh;
        }
    }
}

int orc_batch_linearize_full(const orc_batch_problem* p, const double* poses, double* Hband, double* g, double* cost_out) {
    const int K = p->K, band = p->band;
    double cost = 0;
    if (!orc_batch_linearize(K, band, poses, p->n_con, p->ci, p->cj, p->cp, p->norm_cent, p->score, Hband, g, &cost)) return 0;
    /* attitude constraints: blocks (q_i, q_j), local columns 3..5 of either keyframe */
    for (int f = 0; f < p->n_dq; ++f) {
        const int a = p->dq_i[f], b = p->dq_j[f];
        if (a == b || abs(a - b) > band) return 0;
        const double* P[2] = {poses + 7 * (size_t)a + 3, poses + 7 * (size_t)b + 3};
        double r[3], J0[12], J1[12];
        double* J[2] = {J0, J1};
        orc_eval_delta_q(p->dq_const + 4 * (size_t)f, P, r, J);
        double Pa[12], Pb[12], Ja[18] = {0}, Jb[18] = {0};
        plusJ(P[0], Pa); plusJ(P[1], Pb);
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) {
                Ja[i * 6 + 3 + k] = J0[i * 4] * Pa[k] + J0[i * 4 + 1] * Pa[3 + k] + J0[i * 4 + 2] * Pa[6 + k] + J0[i * 4 + 3] * Pa[9 + k];
                Jb[i * 6 + 3 + k] = J1[i * 4] * Pb[k] + J1[i * 4 + 1] * Pb[3 + k] + J1[i * 4 + 2] * Pb[6 + k] + J1[i * 4 + 3] * Pb[9 + k];
            }
        for (int i = 0; i < 3; ++i) cost += 0.5 * r[i] * r[i];
        band_add(band, Hband, g, a, b, 3, Ja, Jb, r);
    }
    /* double-differenced pseudoranges: blocks (t_i, t_j), local columns 0..2 */
    for (int f = 0; f < p->n_dd; ++f) {
        const glio_dd_psr* F = &p->dd[f];
        const int a = F->slot_i, b = F->slot_j;
        if (a == b || abs(a - b) > band) return 0;
        const double yaw[1] = {p->frame.yaw_enu_local};
        const double* P[4] = {poses + 7 * (size_t)a, poses + 7 * (size_t)b, yaw, p->frame.anc_ecef};
        double r[19], J0[57], J1[57];
        double* J[4] = {J0, J1, NULL, NULL};
        orc_eval_dd_psr(F, P, r, J);
        double Ja[19 * 6] = {0}, Jb[19 * 6] = {0};
        for (int i = 0; i < 19; ++i) for (int k = 0; k < 3; ++k) { Ja[i * 6 + k] = J0[i * 3 + k]; Jb[i * 6 + k] = J1[i * 3 + k]; }
        for (int i = 0; i < 19; ++i) cost += 0.5 * r[i] * r[i];
        band_add(band, Hband, g, a, b, 19, Ja, Jb, r);
    }
    /* LidarPoseFactorBatchRelativeAutoDiff (sms_fusion_level 0, Estimator.cpp:2897-2955): blocks (t_i, q_i, t_j, q_j), all six local columns of either keyframe */
    for (int f = 0; f < p->n_rp; ++f) {
        const int a = p->rp_i[f], b = p->rp_j[f];
        if (a == b || abs(a - b) > band) return 0;
        const double* P[4] = {poses + 7 * (size_t)a, poses + 7 * (size_t)a + 3, poses + 7 * (size_t)b, poses + 7 * (size_t)b + 3};
        double r[6], J0[18], J1[24], J2[18], J3[24];
        double* J[4] = {J0, J1, J2, J3};
        orc_eval_relative_pose(p->rp_const + 7 * (size_t)f, p->rp_const + 7 * (size_t)f + 4, P, r, J);
        double Pa[12], Pb[12], Ja[36], Jb[36];
        plusJ(P[1], Pa); plusJ(P[3], Pb);
        for (int i = 0; i < 6; ++i)
            for (int k = 0; k < 3; ++k) {
                Ja[i * 6 + k] = J0[i * 3 + k]; Jb[i * 6 + k] = J2[i * 3 + k];
                Ja[i * 6 + 3 + k] = J1[i * 4] * Pa[k] + J1[i * 4 + 1] * Pa[3 + k] + J1[i * 4 + 2] * Pa[6 + k] + J1[i * 4 + 3] * Pa[9 + k];
                Jb[i * 6 + 3 + k] = J3[i * 4] * Pb[k] + J3[i * 4 + 1] * Pb[3 + k] + J3[i * 4 + 2] * Pb[6 + k] + J3[i * 4 + 3] * Pb[9 + k];
            }
        for (int i = 0; i < 6; ++i) cost += 0.5 * r[i] * r[i];
        band_add(band, Hband, g, a, b, 6, Ja, Jb, r);
    }
    *cost_out = cost;
    return 1;
}

/* The trust-region minimiser lives in orc_batch2.c (it also covers the IMU chain); the pose-only entry point: */
int orc_batch_solve(const orc_batch_problem* p, const glio_batch_tr_opts* o, double* x, glio_summary* sum) {
    if (p->n_imu != 0) return 0;
    return orc_batch2_solve(p, o, x, NULL, sum, NULL);
}
