/*
 * orc_factors.c -- CPU ORACLE (test infrastructure): the reference's cost functions restated in
 * plain C with the ceres::CostFunction::Evaluate pointer convention (residuals + GLOBAL row-major
 * Jacobians; jacobians or jacobians[i] may be NULL).  See glio_oracle.h for the parity
 * statement.  Each function cites the reference lines it follows.
 */
#include "glio_oracle.h"
#include "orc_math.h"

void orc_opts_default(glio_opts* o) {
    memset(o, 0, sizeof *o);
    o->window = 5;                 /* config_urban_hk.yaml:66 */
    o->max_iterations = 15;        /* Estimator.cpp:2427 */
    o->max_points_per_scan = 65536;
    o->max_map_points = 1 << 21;
    o->max_ddt_epochs = 0;
    o->jacobi_scaling = 1;
    o->huber_delta = 1.0;          /* Estimator.cpp:70 */
    o->doppler_huber_delta = 1.0;  /* Estimator.cpp:2335 */
    o->q_lb[0] = 1.0;              /* yaml:90-93 */
    o->t_lb[2] = 0.28;             /* yaml:95-97 */
    o->lidar_const = 7.5;          /* yaml:70 */
    o->surf_dist_thres = 0.18;     /* yaml:71 */
    o->kd_max_radius = 1.5;       /* yaml:72 */
    o->weight_gate = 0.3;         /* Estimator.cpp:3681 */
    o->gravity = 9.80511;          /* yaml:11 */
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
}

/* ------------------------------------------------------------------------------------------------
 * LidarPlaneNormFactor::operator() (LidarKeyframeFactor.h:87-103), differentiated the way
 * ceres::AutoDiffCostFunction<LidarPlaneNormFactor,1,3,4> does: through Eigen's
 * _transformVector formula, giving a 1x3 and a 1x4 GLOBAL Jacobian.
 */
int orc_eval_lidar_plane(const glio_opts* o, const float cp[4], const float plane[4], double score,
                         double const* const* P, double* res, double** J) {
    const double* t = P[0];
    const double* q = P[1];
    double c[3] = {(double)cp[0] - o->t_lb[0], (double)cp[1] - o->t_lb[1], (double)cp[2] - o->t_lb[2]};
    double qlb_inv[4], pb[3], pw[3];
    q_inv(o->q_lb, qlb_inv);
    q_rot(qlb_inv, c, pb);                 /* :97 point_w = q_l_b.inverse() * (cp - t_l_b) */
    q_rot(q, pb, pw);                      /* :98 */
    pw[0] += t[0]; pw[1] += t[1]; pw[2] += t[2];
    double n[3] = {(double)plane[0], (double)plane[1], (double)plane[2]};
    double d = (double)plane[3];
    res[0] = score * (v3_dot(n, pw) + d);  /* :101 */
    if (J) {
        if (J[0]) { J[0][0] = score * n[0]; J[0][1] = score * n[1]; J[0][2] = score * n[2]; }
        if (J[1]) {
            /* f(w,u) = v + 2w(u x v) + 2 u x (u x v);  df/dw = 2(u x v);
             * df/du = -2w[v]x - 2[u x v]x - 2[u]x[v]x */
            const double* u = q + 1;
            double w = q[0];
            double uv[3];
            v3_cross(u, pb, uv);
            double Sv[9], Suv[9], Su[9], SuSv[9];
            skew3(pb, Sv); skew3(uv, Suv); skew3(u, Su);
            mat_mul(Su, Sv, SuSv, 3, 3, 3);
            double dfdw[3] = {2 * uv[0], 2 * uv[1], 2 * uv[2]};
            J[1][0] = score * v3_dot(n, dfdw);
            for (int k = 0; k < 3; ++k) {
                double col[3];
                for (int r = 0; r < 3; ++r) col[r] = -2 * w * Sv[r * 3 + k] - 2 * Suv[r * 3 + k] - 2 * SuSv[r * 3 + k];
                J[1][1 + k] = score * v3_dot(n, col);
            }
        }
    }
    return 1;
}

/* sqrt_info = LLT(covariance^-1).matrixL().transpose()   (ImuFactor.h:44-45) */
int orc_imu_sqrt_info(const double* cov, double* sqrt_info) {
    double A[225], Ainv[225];
    memcpy(A, cov, sizeof A);
    if (mat_inverse(A, Ainv, 15)) return 0;
    /* symmetrise the tiny asymmetry of the numerical inverse the way LLT reads it: lower triangle */
    if (chol_lower(Ainv, 15)) return 0;
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) sqrt_info[i * 15 + j] = (j >= i) ? Ainv[j * 15 + i] : 0.0;
    return 1;
}

static void left_mul_15(const double* S, double* M, int cols) {
    double tmp[15 * 9];
    mat_mul(S, M, tmp, 15, 15, cols);
    memcpy(M, tmp, sizeof(double) * 15 * cols);
}

/* ImuFactor::Evaluate (ImuFactor.h:21-171) with Preintegration::evaluate (Preintegration.h:196-235) */
int orc_eval_imu(const glio_opts* o, const glio_preint* pre, double const* const* P, double* res, double** J) {
    enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };
    const double* Pi = P[0];
    double Qi[4] = {P[1][0], P[1][1], P[1][2], P[1][3]};
    q_normalize(Qi);                                                  /* :25 */
    const double* Vi = P[2];
    const double* Bai = P[2] + 3;
    const double* Bgi = P[2] + 6;
    const double* Pj = P[3];
    double Qj[4] = {P[4][0], P[4][1], P[4][2], P[4][3]};
    q_normalize(Qj);                                                  /* :33 */
    const double* Vj = P[5];
    const double* Baj = P[5] + 3;
    const double* Bgj = P[5] + 6;
    const double g[3] = {0, 0, -o->gravity};                          /* Preintegration.h:58 */
    const double dt = pre->sum_dt;

    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            dp_dba[r * 3 + c] = pre->jacobian[(O_P + r) * 15 + O_BA + c];
            dp_dbg[r * 3 + c] = pre->jacobian[(O_P + r) * 15 + O_BG + c];
            dq_dbg[r * 3 + c] = pre->jacobian[(O_R + r) * 15 + O_BG + c];
            dv_dba[r * 3 + c] = pre->jacobian[(O_V + r) * 15 + O_BA + c];
            dv_dbg[r * 3 + c] = pre->jacobian[(O_V + r) * 15 + O_BG + c];
        }
    double dba[3], dbg[3];
    for (int k = 0; k < 3; ++k) { dba[k] = Bai[k] - pre->linearized_ba[k]; dbg[k] = Bgi[k] - pre->linearized_bg[k]; }

    /* Preintegration.h:223-225 */
    double th[3], dq[4], cdq[4];
    mat3_vec(dq_dbg, dbg, th);
    delta_q(th, dq);
    q_mul(pre->delta_q, dq, cdq);
    double cdv[3], cdp[3], t1[3], t2[3];
    mat3_vec(dv_dba, dba, t1); mat3_vec(dv_dbg, dbg, t2);
    for (int k = 0; k < 3; ++k) cdv[k] = pre->delta_v[k] + t1[k] + t2[k];
    mat3_vec(dp_dba, dba, t1); mat3_vec(dp_dbg, dbg, t2);
    for (int k = 0; k < 3; ++k) cdp[k] = pre->delta_p[k] + t1[k] + t2[k];

    double Qi_inv[4], tmp[3], tmp1[3], rot[3];
    q_inv(Qi, Qi_inv);
    for (int k = 0; k < 3; ++k) {
        tmp[k] = -0.5 * g[k] * dt * dt + Pj[k] - Pi[k] - Vi[k] * dt;
        tmp1[k] = -g[k] * dt + Vj[k] - Vi[k];
    }
    double r[15];
    q_rot(Qi_inv, tmp, rot);                                          /* :227-228 */
    for (int k = 0; k < 3; ++k) r[O_P + k] = rot[k] - cdp[k];
    double cdq_inv[4], qij[4], qe[4];
    q_inv(cdq, cdq_inv);
    q_mul(Qi_inv, Qj, qij);
    q_mul(cdq_inv, qij, qe);
    q_normalize(qe);                                                  /* :229 */
    for (int k = 0; k < 3; ++k) r[O_R + k] = 2.0 * qe[1 + k];
    q_rot(Qi_inv, tmp1, rot);                                         /* :230 */
    for (int k = 0; k < 3; ++k) r[O_V + k] = rot[k] - cdv[k];
    for (int k = 0; k < 3; ++k) { r[O_BA + k] = Baj[k] - Bai[k]; r[O_BG + k] = Bgj[k] - Bgi[k]; }

    double S[225];
    if (!orc_imu_sqrt_info(pre->covariance, S)) return 0;
    mat_mul(S, r, res, 15, 15, 1);                                    /* ImuFactor.h:47 */

    if (!J) return 1;
    double Ri_inv[9];
    q_to_R(Qi_inv, Ri_inv);
    const double w = Qi[0];
    const double* u = Qi + 1;

    if (J[0]) {                                                       /* :63-74 */
        double* M = J[0];
        memset(M, 0, sizeof(double) * 15 * 3);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[(O_P + a) * 3 + b] = -Ri_inv[a * 3 + b];
        left_mul_15(S, M, 3);
    }
    if (J[1]) {                                                       /* :77-98 */
        double* M = J[1];
        memset(M, 0, sizeof(double) * 15 * 4);
        const double* vv[2] = {tmp, tmp1};
        const int rows[2] = {O_P, O_V};
        for (int s = 0; s < 2; ++s) {
            const double* v = vv[s];
            double uxv[3], Sv[9];
            v3_cross(u, v, uxv);            /* skewSymmetric(Qi.vec()) * tmp */
            skew3(v, Sv);
            double udv = v3_dot(u, v);
            for (int a = 0; a < 3; ++a) {
                M[(rows[s] + a) * 4 + 0] = 2 * (w * v[a] + uxv[a]);
                for (int b = 0; b < 3; ++b)
                    M[(rows[s] + a) * 4 + 1 + b] =
                        2 * ((a == b ? udv : 0.0) + u[a] * v[b] - v[a] * u[b] - w * Sv[a * 3 + b]);
            }
        }
        double Qj_inv[4], L[16], R[16], LR[16];
        q_inv(Qj, Qj_inv);
        q_left(Qj_inv, L);
        q_right(cdq, R);
        mat_mul(L, R, LR, 4, 4, 4);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 4; ++b) M[(O_R + a) * 4 + b] = -2 * LR[(1 + a) * 4 + b];
        left_mul_15(S, M, 4);
    }
    if (J[2]) {                                                       /* :102-124 */
        double* M = J[2];
        memset(M, 0, sizeof(double) * 15 * 9);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                M[(O_P + a) * 9 + 0 + b] = -Ri_inv[a * 3 + b] * dt;
                M[(O_P + a) * 9 + 3 + b] = -dp_dba[a * 3 + b];
                M[(O_P + a) * 9 + 6 + b] = -dp_dbg[a * 3 + b];
                M[(O_V + a) * 9 + 0 + b] = -Ri_inv[a * 3 + b];
                M[(O_V + a) * 9 + 3 + b] = -dv_dba[a * 3 + b];
                M[(O_V + a) * 9 + 6 + b] = -dv_dbg[a * 3 + b];
            }
        double Qj_inv[4], qa[4], qb[4];
        q_inv(Qj, Qj_inv);
        q_mul(Qj_inv, Qi, qa);
        q_mul(qa, cdq, qb);
        /* LeftQuatMatrix(qb).topLeftCorner<3,3>() = w I + [vec]x   (math_tools.h:141-150) */
        double Sx[9], TL[9], prod[9];
        skew3(qb + 1, Sx);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) TL[a * 3 + b] = (a == b ? qb[0] : 0.0) + Sx[a * 3 + b];
        mat_mul(TL, dq_dbg, prod, 3, 3, 3);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[(O_R + a) * 9 + 6 + b] = -prod[a * 3 + b];
        for (int a = 0; a < 3; ++a) { M[(O_BA + a) * 9 + 3 + a] = -1.0; M[(O_BG + a) * 9 + 6 + a] = -1.0; }
        left_mul_15(S, M, 9);
    }
    if (J[3]) {                                                       /* :127-135 */
        double* M = J[3];
        memset(M, 0, sizeof(double) * 15 * 3);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[(O_P + a) * 3 + b] = Ri_inv[a * 3 + b];
        left_mul_15(S, M, 3);
    }
    if (J[4]) {                                                       /* :139-150 */
        double* M = J[4];
        memset(M, 0, sizeof(double) * 15 * 4);
        double qa[4], L[16];
        q_mul(cdq_inv, Qi_inv, qa);
        q_left(qa, L);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 4; ++b) M[(O_R + a) * 4 + b] = 2 * L[(1 + a) * 4 + b];
        left_mul_15(S, M, 4);
    }
    if (J[5]) {                                                       /* :155-167 */
        double* M = J[5];
        memset(M, 0, sizeof(double) * 15 * 9);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[(O_V + a) * 9 + b] = Ri_inv[a * 3 + b];
        for (int a = 0; a < 3; ++a) { M[(O_BA + a) * 9 + 3 + a] = 1.0; M[(O_BG + a) * 9 + 6 + a] = 1.0; }
        left_mul_15(S, M, 9);
    }
    return 1;
}

static int blk_size(int kind) { return kind == GLIO_BLK_TRANS ? 3 : (kind == GLIO_BLK_QUAT ? 4 : 9); }

/* MarginalizationFactor::Evaluate (MarginalizationFactor.cpp:233-287) */
int orc_eval_marg(const glio_prior* p, double const* const* P, double* res, double** J) {
    const int n = p->n;
    double dx[GLIO_MAX_WINDOW * 6 + 9];
    for (int b = 0; b < p->n_blocks; ++b) {
        const int size = blk_size(p->blk_kind[b]);
        const int idx = p->blk_idx[b];
        const double* x = P[b];
        const double* x0 = p->blk_x0 + 9 * b;
        if (size != 4) {
            for (int k = 0; k < size; ++k) dx[idx + k] = x[k] - x0[k];            /* :243 */
        } else {
            double q0inv[4], dq[4];
            q_inv(x0, q0inv);
            q_mul(q0inv, x, dq);
            double wsign = dq[0];
            q_normalize(dq);
            for (int k = 0; k < 3; ++k) dx[idx + k] = (wsign < 0 ? -2.0 : 2.0) * dq[1 + k];  /* :246-252 */
        }
    }
    for (int i = 0; i < n; ++i) {                                                 /* :256-257 */
        double s = p->lin_res[i];
        for (int k = 0; k < n; ++k) s += p->lin_jac[i * n + k] * dx[k];
        res[i] = s;
    }
    if (!J) return 1;
    for (int b = 0; b < p->n_blocks; ++b) {
        if (!J[b]) continue;
        const int size = blk_size(p->blk_kind[b]);
        const int idx = p->blk_idx[b];
        double* M = J[b];
        if (size != 4) {
            for (int i = 0; i < n; ++i) for (int k = 0; k < size; ++k) M[i * size + k] = p->lin_jac[i * n + idx + k];
        } else {
            const double* x = P[b];
            const double* x0 = p->blk_x0 + 9 * b;
            double q0inv[4], dq[4], L[16];
            q_inv(x0, q0inv);
            q_mul(q0inv, x, dq);
            const double s = (dq[0] >= 0) ? 2.0 : -2.0;                           /* :276-281 */
            q_left(q0inv, L);
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < 4; ++c) {
                    double a = 0;
                    for (int k = 0; k < 3; ++k) a += p->lin_jac[i * n + idx + k] * L[(1 + k) * 4 + c];
                    M[i * 4 + c] = s * a;
                }
        }
    }
    return 1;
}

/* ecef2geo + geo2rotation  (gnss_comm/src/gnss_utility.cpp:347-390,738-748) */
void orc_ecef2rotation(const double xyz[3], double R[9]) {
    const double e2 = 6.69437999014e-3;   /* EARTH_ECCE_2, gnss_constant.hpp:214 */
    const double a = 6378137.0;           /* EARTH_SEMI_MAJOR, :216 */
    const double R2D = 180.0 / M_PI, D2R = M_PI / 180.0;
    double a2 = a * a, b2 = a2 * (1 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2;
    double p = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
    double s1 = xyz[2] * a, s2 = p * b, h = sqrt(s1 * s1 + s2 * s2);
    double sin_theta = s1 / h, cos_theta = s2 / h;
    s1 = xyz[2] + ep2 * b * pow(sin_theta, 3);
    s2 = p - a * e2 * pow(cos_theta, 3);
    double tan_lat = s1 / s2;
    double lat_deg = atan(tan_lat) * R2D;
    double lon_deg = atan2(xyz[1], xyz[0]) * R2D;
    double lat = lat_deg * D2R, lon = lon_deg * D2R;
    double sin_lat = sin(lat), cos_lat = cos(lat), sin_lon = sin(lon), cos_lon = cos(lon);
    R[0] = -sin_lon; R[1] = -sin_lat * cos_lon; R[2] = cos_lat * cos_lon;
    R[3] = cos_lon;  R[4] = -sin_lat * sin_lon; R[5] = cos_lat * sin_lon;
    R[6] = 0;        R[7] = cos_lat;            R[8] = sin_lat;
}

/* dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171) */
int orc_eval_dd_psr(const glio_dd_psr* f, double const* const* P, double* res, double** J) {
    enum { NR = GLIO_DD_MAX_SAT - 1 };
    const double* Pi = P[0];
    const double* Pj = P[1];
    const double yaw = P[2][0];
    const double* anc = P[3];
    double s = sin(yaw), c = cos(yaw);
    double Rel[9] = {c, -s, 0, s, c, 0, 0, 0, 1};                    /* :35-38 */
    double Ree[9], R[9];
    orc_ecef2rotation(anc, Ree);
    mat_mul(Ree, Rel, R, 3, 3, 3);                                    /* :40 */
    double lp[3], Pe[3];
    for (int k = 0; k < 3; ++k) lp[k] = f->ratio * Pi[k] + (1.0 - f->ratio) * Pj[k];  /* :42-43, lever arm zero */
    mat3_vec(R, lp, Pe);
    for (int k = 0; k < 3; ++k) Pe[k] += anc[k];                      /* :45 */
    double raw[NR], Ji[NR * 3], Jj[NR * 3];
    memset(raw, 0, sizeof raw); memset(Ji, 0, sizeof Ji); memset(Jj, 0, sizeof Jj);
    const int m = f->master, ns = f->n_sat;
    int ri = 0;
    for (int i = 0; i < ns; ++i) {
        if (i == m) continue;
        double d_ui[3], d_um[3], d_ri[3], d_rm[3];
        for (int k = 0; k < 3; ++k) {
            d_ui[k] = f->user_sat_pos[i][k] - Pe[k];                  /* :75 */
            d_um[k] = f->user_sat_pos[m][k] - Pe[k];                  /* :79 */
            d_ri[k] = f->ref_sat_pos[i][k] - f->station[k];           /* :83 */
            d_rm[k] = f->ref_sat_pos[m][k] - f->station[k];           /* :87 */
        }
        double r_ui = v3_norm(d_ui), r_um = v3_norm(d_um), r_ri = v3_norm(d_ri), r_rm = v3_norm(d_rm);
        double est = (r_ui - r_ri) - (r_um - r_rm);                   /* :95 */
        double obs = (f->user_psr[i] - f->ref_psr[i]) - (f->user_psr[m] - f->ref_psr[m]);  /* :97 */
        double wgt = 1.0;
        if (fabs(est - obs) > f->threshold) wgt = 0.05;               /* :99-102 */
        raw[ri] = wgt * (est - obs);
        /* :111,118  J = (-e_i^T R + e_m^T R) * w * ratio */
        double e_i[3] = {d_ui[0] / r_ui, d_ui[1] / r_ui, d_ui[2] / r_ui};
        double e_m[3] = {d_um[0] / r_um, d_um[1] / r_um, d_um[2] / r_um};
        for (int cidx = 0; cidx < 3; ++cidx) {
            double ei = e_i[0] * R[0 * 3 + cidx] + e_i[1] * R[1 * 3 + cidx] + e_i[2] * R[2 * 3 + cidx];
            double em = e_m[0] * R[0 * 3 + cidx] + e_m[1] * R[1 * 3 + cidx] + e_m[2] * R[2 * 3 + cidx];
            Ji[ri * 3 + cidx] = (-ei * wgt * f->ratio) - (-em * wgt * f->ratio);
            Jj[ri * 3 + cidx] = (-ei * wgt * (1.0 - f->ratio)) - (-em * wgt * (1.0 - f->ratio));
        }
        ++ri;
    }
    /* :57-59,151-167  residual = W_ep * residual, J = W_ep * J, W embedded top-left */
    const int nw = ns - 1;
    for (int a = 0; a < NR; ++a) {
        double sr = 0, si[3] = {0, 0, 0}, sj[3] = {0, 0, 0};
        if (a < nw)
            for (int b = 0; b < nw; ++b) {
                double wv = f->weight[a * nw + b];
                sr += wv * raw[b];
                for (int k = 0; k < 3; ++k) { si[k] += wv * Ji[b * 3 + k]; sj[k] += wv * Jj[b * 3 + k]; }
            }
        res[a] = sr;
        if (J && J[0]) for (int k = 0; k < 3; ++k) J[0][a * 3 + k] = si[k];
        if (J && J[1]) for (int k = 0; k < 3; ++k) J[1][a * 3 + k] = sj[k];
    }
    return 1;
}

/* tcdopplerFactor::operator() (dopp_factor.hpp:24-75), differentiated analytically (the reference
 * uses ceres::AutoDiffCostFunction<tcdopplerFactor,1,3,9,3,9,EPOCH_SIZE,1,3>, Estimator.cpp:2329-2331) */
int orc_eval_doppler(const glio_doppler* f, double const* const* P, double* res, double** J) {
    const double OMG = 7.2921151467e-5;   /* EARTH_OMG_GPS gnss_constant.hpp:219 */
    const double CLIGHT = 2.99792458e8;   /* LIGHT_SPEED :225 */
    const double* Pi = P[0];
    const double* Vi = P[1];
    const double* Pj = P[2];
    const double* Vj = P[3];
    const double rcv_ddt = P[4][f->epoch];                            /* :38 */
    const double* anc = P[6];
    const double* R = f->R_ecef_local;
    double lp[3], lv[3], Pe[3], Ve[3];
    for (int k = 0; k < 3; ++k) {
        lp[k] = f->ratio * Pi[k] + (1.0 - f->ratio) * Pj[k] + f->lever_arm[k];   /* :52-53 */
        lv[k] = f->ratio * Vi[k] + (1.0 - f->ratio) * Vj[k];                      /* :54 */
    }
    mat3_vec(R, lp, Pe); mat3_vec(R, lv, Ve);
    for (int k = 0; k < 3; ++k) Pe[k] += anc[k];                      /* :57 */
    double d[3] = {f->sat_pos[0] - Pe[0], f->sat_pos[1] - Pe[1], f->sat_pos[2] - Pe[2]};
    double rho = v3_norm(d);
    double e[3] = {d[0] / rho, d[1] / rho, d[2] / rho};               /* :61-62 */
    double sag = OMG / CLIGHT * (f->sat_vel[0] * Pe[1] + f->sat_pos[0] * Ve[1]
                                 - f->sat_vel[1] * Pe[0] - f->sat_pos[1] * Ve[0]);   /* :65-66 */
    double a[3] = {f->sat_vel[0] - Ve[0], f->sat_vel[1] - Ve[1], f->sat_vel[2] - Ve[2]};
    double ae = v3_dot(a, e);
    double est = ae + sag + rcv_ddt - f->sv_ddt;                      /* :69 */
    res[0] = (est + f->doppler * f->lamda) / f->var;                  /* :72 */
    if (!J) return 1;
    /* d/dPe = -(a - (a.e) e)/rho + OMG/c (-svel_y, svel_x, 0);  d/dVe = -e + OMG/c (-spos_y, spos_x, 0) */
    double gP[3], gV[3];
    for (int k = 0; k < 3; ++k) { gP[k] = -(a[k] - ae * e[k]) / rho; gV[k] = -e[k]; }
    gP[0] += OMG / CLIGHT * (-f->sat_vel[1]); gP[1] += OMG / CLIGHT * f->sat_vel[0];
    gV[0] += OMG / CLIGHT * (-f->sat_pos[1]); gV[1] += OMG / CLIGHT * f->sat_pos[0];
    double gPl[3], gVl[3];   /* row vectors times R */
    for (int c = 0; c < 3; ++c) {
        gPl[c] = gP[0] * R[c] + gP[1] * R[3 + c] + gP[2] * R[6 + c];
        gVl[c] = gV[0] * R[c] + gV[1] * R[3 + c] + gV[2] * R[6 + c];
    }
    const double iv = 1.0 / f->var;
    if (J[0]) for (int k = 0; k < 3; ++k) J[0][k] = f->ratio * gPl[k] * iv;
    if (J[1]) { for (int k = 0; k < 9; ++k) J[1][k] = 0; for (int k = 0; k < 3; ++k) J[1][k] = f->ratio * gVl[k] * iv; }
    if (J[2]) for (int k = 0; k < 3; ++k) J[2][k] = (1.0 - f->ratio) * gPl[k] * iv;
    if (J[3]) { for (int k = 0; k < 9; ++k) J[3][k] = 0; for (int k = 0; k < 3; ++k) J[3][k] = (1.0 - f->ratio) * gVl[k] * iv; }
    if (J[4]) J[4][0] = iv;
    return 1;
}

/* BinaryLidarPlaneNormFactor::operator() (LidarKeyframeFactor.h:132-150): no extrinsic (Q10) */
int orc_eval_binary_plane(const float cp[4], const double pnc[6], double score,
                          double const* const* P, double* res, double** J) {
    const double* t1 = P[0]; const double* q1 = P[1]; const double* t2 = P[2]; const double* q2 = P[3];
    double p[3] = {(double)cp[0], (double)cp[1], (double)cp[2]};
    double pw[3], no[3], co[3];
    q_rot(q1, p, pw);
    for (int k = 0; k < 3; ++k) pw[k] += t1[k];                        /* :144 */
    q_rot(q2, pnc, no);                                                /* :145 */
    q_rot(q2, pnc + 3, co);
    for (int k = 0; k < 3; ++k) co[k] += t2[k];                        /* :146 */
    double diff[3] = {pw[0] - co[0], pw[1] - co[1], pw[2] - co[2]};
    res[0] = score * v3_dot(no, diff);                                 /* :148 */
    if (!J) return 1;
    if (J[0]) for (int k = 0; k < 3; ++k) J[0][k] = score * no[k];
    if (J[2]) for (int k = 0; k < 3; ++k) J[2][k] = -score * no[k];
    /* d(q*v)/dq through Eigen's _transformVector (see orc_eval_lidar_plane) */
    for (int which = 0; which < 2; ++which) {
        double* Jq = which == 0 ? J[1] : J[3];
        if (!Jq) continue;
        const double* q = which == 0 ? q1 : q2;
        const double* u = q + 1;
        const double w = q[0];
        /* for q1: r = s * no . f(q1,p)            -> row = s*no, v = p
         * for q2: r = s * (f(q2,n_l) . diff - no . f(q2,c_l)) */
        double acc[4] = {0, 0, 0, 0};
        const int nterms = which == 0 ? 1 : 2;
        for (int term = 0; term < nterms; ++term) {
            const double* v; double row[3];
            if (which == 0) { v = p; for (int k = 0; k < 3; ++k) row[k] = score * no[k]; }
            else if (term == 0) { v = pnc; for (int k = 0; k < 3; ++k) row[k] = score * diff[k]; }
            else { v = pnc + 3; for (int k = 0; k < 3; ++k) row[k] = -score * no[k]; }
            double uv[3], Sv[9], Suv[9], Su[9], SuSv[9];
            v3_cross(u, v, uv);
            skew3(v, Sv); skew3(uv, Suv); skew3(u, Su);
            mat_mul(Su, Sv, SuSv, 3, 3, 3);
            acc[0] += 2 * v3_dot(row, uv);
            for (int k = 0; k < 3; ++k) {
                double col[3];
                for (int r = 0; r < 3; ++r) col[r] = -2 * w * Sv[r * 3 + k] - 2 * Suv[r * 3 + k] - 2 * SuSv[r * 3 + k];
                acc[1 + k] += v3_dot(row, col);
            }
        }
        for (int k = 0; k < 4; ++k) Jq[k] = acc[k];
    }
    return 1;
}
