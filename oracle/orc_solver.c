/*
 * orc_solver.c -- CPU ORACLE (test infrastructure): the sliding-window Problem of
 * Estimator::optimizeSlidingWindowWithLandMark (GLIO/src/Estimator.cpp:2091-2433) assembled and
 * minimised with Ceres-1.14 semantics:
 *   - residual blocks in the order the reference adds them (prior :2153-2158, IMU :2182-2192,
 *     LiDAR planes :2198-2248, Doppler + DD pseudorange :2255-2421 as defined, F4)
 *   - loss corrector (nnls_modeling.rst:1146-1186, concrete spec MarginalizationFactor.cpp:44-70)
 *   - QuaternionParameterization plus / Jacobian (nnls_modeling.rst:1312-1327)
 *   - Jacobi scaling, traditional dogleg, monotonic steps, Ceres default tolerances
 *     (nnls_solving.rst:217-260,1056-1188,1402-1404; Estimator.cpp:2424-2430)
 * Dogleg radius-update constants (decrease 0.25 / increase 0.75 / x0.5 / max(r, 3|step|), mu 1e-8..1,
 * x10) are those of the public Ceres 1.14 dogleg_strategy.cc; they are not in the bundled docs.
 * PARITY UNPINNED -- see glio_oracle.h.
 */
#include <stdlib.h>
#include "glio_oracle.h"
#include "orc_math.h"

/* d([1,delta] (x) q)/d delta at 0: 4x3 row-major  (Ceres QuaternionParameterization::ComputeJacobian) */
static void quat_plus_jacobian(const double q[4], double P[12]) {
    P[0] = -q[1]; P[1] = -q[2]; P[2] = -q[3];
    P[3] = q[0];  P[4] = q[3];  P[5] = -q[2];
    P[6] = -q[3]; P[7] = q[0];  P[8] = q[1];
    P[9] = q[2];  P[10] = -q[1]; P[11] = q[0];
}

void orc_quat_plus(const double q[4], const double d[3], double out[4]) {
    const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nrm > 0.0) {
        const double s = sin(nrm) / nrm;
        double dq[4] = {cos(nrm), s * d[0], s * d[1], s * d[2]};
        q_mul(dq, q, out);
    } else {
        memcpy(out, q, 4 * sizeof(double));
    }
}

void orc_state_plus(const glio_state* x, int W, const double* delta, glio_state* out) {
    for (int s = 0; s < W; ++s) {
        const double* d = delta + 15 * s;
        for (int k = 0; k < 3; ++k) out->trans[3 * s + k] = x->trans[3 * s + k] + d[k];
        orc_quat_plus(x->quat + 4 * s, d + 3, out->quat + 4 * s);
        for (int k = 0; k < 9; ++k) out->speed_bias[9 * s + k] = x->speed_bias[9 * s + k] + d[6 + k];
    }
    for (int e = 0; e < x->n_ddt; ++e) out->rcv_ddt[e] = x->rcv_ddt[e] + delta[15 * W + e];
}

int orc_local_dim(const orc_problem* p, const glio_state* x) { return 15 * p->opts.window + x->n_ddt; }

/* Huber (Ceres HuberLoss::Evaluate) */
static void huber(double a, double s, double rho[3]) {
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r;
        if (rho[1] < 2.2250738585072014e-308) rho[1] = 2.2250738585072014e-308;
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

/* A residual block after evaluation: nr residuals, nb parameter blocks with GLOBAL Jacobians. */
typedef struct {
    int nr, nb;
    double* r;
    double* Jg[8];       /* nr x gsize[b] row-major */
    int gsize[8];
    int lsize[8];
    int loff[8];         /* offset in the local state vector, -1 = constant block */
    const double* quat[8];  /* current quaternion for 4-blocks */
} rblock;

/* loss correction (MarginalizationFactor.cpp:44-70) then local parameterisation then H += J^T J */
static void accumulate(const rblock* b, int use_loss, double loss_a, int n, double* H, double* g, double* cost) {
    double sq = 0;
    for (int i = 0; i < b->nr; ++i) sq += b->r[i] * b->r[i];
    double rscale = 1.0, sqrt_rho1 = 1.0, alpha_sq = 0.0, c = 0.5 * sq;
    if (use_loss) {
        double rho[3];
        huber(loss_a, sq, rho);
        c = 0.5 * rho[0];
        sqrt_rho1 = sqrt(rho[1]);
        if (sq == 0.0 || rho[2] <= 0.0) {
            rscale = sqrt_rho1; alpha_sq = 0.0;
        } else {
            const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
            const double alpha = 1.0 - sqrt(D);
            rscale = sqrt_rho1 / (1 - alpha);
            alpha_sq = alpha / sq;
        }
    }
    *cost += c;
    if (!H && !g) return;
    /* local Jacobian, nr x ltot, with column -> global-state index map */
    int ltot = 0, cols[64];
    for (int k = 0; k < b->nb; ++k) if (b->loff[k] >= 0) for (int j = 0; j < b->lsize[k]; ++j) cols[ltot++] = b->loff[k] + j;
    double* Jl = (double*)malloc(sizeof(double) * b->nr * (ltot > 0 ? ltot : 1));
    int c0 = 0;
    for (int k = 0; k < b->nb; ++k) {
        if (b->loff[k] < 0) continue;
        const int gs = b->gsize[k], ls = b->lsize[k];
        /* corrected global jacobian row i: sqrt_rho1 * (J - alpha_sq * r (r^T J)) */
        double rtJ[9];
        for (int j = 0; j < gs; ++j) { double s = 0; for (int i = 0; i < b->nr; ++i) s += b->r[i] * b->Jg[k][i * gs + j]; rtJ[j] = s; }
        double Pj[12] = {0};
        if (gs == 4) quat_plus_jacobian(b->quat[k], Pj);
        for (int i = 0; i < b->nr; ++i) {
            double row[9];
            for (int j = 0; j < gs; ++j) row[j] = sqrt_rho1 * (b->Jg[k][i * gs + j] - alpha_sq * b->r[i] * rtJ[j]);
            if (gs == 4) {
                for (int j = 0; j < 3; ++j)
                    Jl[i * ltot + c0 + j] = row[0] * Pj[0 + j] + row[1] * Pj[3 + j] + row[2] * Pj[6 + j] + row[3] * Pj[9 + j];
            } else {
                for (int j = 0; j < ls; ++j) Jl[i * ltot + c0 + j] = row[j];
            }
        }
        c0 += ls;
    }
    for (int a = 0; a < ltot; ++a) {
        double gr = 0;
        for (int i = 0; i < b->nr; ++i) gr += Jl[i * ltot + a] * (b->r[i] * rscale);
        if (g) g[cols[a]] += gr;
        if (H)
            for (int bb = 0; bb < ltot; ++bb) {
                double s = 0;
                for (int i = 0; i < b->nr; ++i) s += Jl[i * ltot + a] * Jl[i * ltot + bb];
                H[(size_t)cols[a] * n + cols[bb]] += s;
            }
    }
    free(Jl);
}

/* threads of the all-cores baseline variant (1 = the reference's options.num_threads = 1, Estimator.cpp:2426: the parity path) */
static int g_orc_threads = 1;
void orc_set_threads(int t) { g_orc_threads = t < 1 ? 1 : t; }

/* the LidarPlaneNormFactor blocks of keyframe s (Estimator.cpp:2226-2242), cost added residual by residual into *cost */
static void lidar_keyframe(const orc_problem* p, const glio_state* x, int s, int n, double* H, double* g, double* cost, int want_J) {
    const double* P[2] = {x->trans + 3 * s, x->quat + 4 * s};
    double Pq[12];
    quat_plus_jacobian(P[1], Pq);
    for (int i = p->lidar_offset[s]; i < p->lidar_offset[s + 1]; ++i) {
        double r, Jt[3], Jq[4];
        double* J[2] = {Jt, Jq};
        orc_eval_lidar_plane(&p->opts, p->lidar_pts + 4 * (size_t)i, p->lidar_planes + 4 * (size_t)i,
                             p->lidar_scores[i], P, &r, want_J ? J : NULL);
        /* scalar residual: Huber has rho'' <= 0 so the corrector reduces to sqrt(rho') scaling */
        double rho[3];
        huber(p->opts.huber_delta, r * r, rho);
        *cost += 0.5 * rho[0];
        if (!want_J) continue;
        const double sr = sqrt(rho[1]);
        double Jl[6];
        for (int k = 0; k < 3; ++k) Jl[k] = sr * Jt[k];
        for (int k = 0; k < 3; ++k)
            Jl[3 + k] = sr * (Jq[0] * Pq[k] + Jq[1] * Pq[3 + k] + Jq[2] * Pq[6 + k] + Jq[3] * Pq[9 + k]);
        const double rc = sr * r;
        const int o = 15 * s;
        for (int a = 0; a < 6; ++a) {
            if (g) g[o + a] += Jl[a] * rc;
            if (H) for (int bb = 0; bb < 6; ++bb) H[(size_t)(o + a) * n + o + bb] += Jl[a] * Jl[bb];
        }
    }
}

int orc_linearize(const orc_problem* p, const glio_state* x, double* H, double* g, double* cost_out) {
    const int W = p->opts.window;
    const int n = 15 * W + x->n_ddt;
    if (H) memset(H, 0, sizeof(double) * (size_t)n * n);
    if (g) memset(g, 0, sizeof(double) * n);
    double cost = 0;
    const int want_J = (H || g);

    /* 1. marginalization prior (Estimator.cpp:2153-2158), no loss */
    if (p->prior.n > 0) {
        const glio_prior* pr = &p->prior;
        const int nn = pr->n, nb = pr->n_blocks;
        const double** P = (const double**)malloc(sizeof(double*) * nb);
        double** J = (double**)malloc(sizeof(double*) * nb);
        double* r = (double*)malloc(sizeof(double) * nn);
        int* goff = (int*)malloc(sizeof(int) * nb);
        for (int b = 0; b < nb; ++b) {
            const int s = pr->blk_slot[b];
            const int kind = pr->blk_kind[b];
            P[b] = kind == GLIO_BLK_TRANS ? x->trans + 3 * s : (kind == GLIO_BLK_QUAT ? x->quat + 4 * s : x->speed_bias + 9 * s);
            const int gs = kind == GLIO_BLK_TRANS ? 3 : (kind == GLIO_BLK_QUAT ? 4 : 9);
            J[b] = want_J ? (double*)malloc(sizeof(double) * nn * gs) : NULL;
            goff[b] = 15 * s + (kind == GLIO_BLK_TRANS ? 0 : (kind == GLIO_BLK_QUAT ? 3 : 6));
        }
        orc_eval_marg(pr, P, r, want_J ? J : NULL);
        double sq = 0;
        for (int i = 0; i < nn; ++i) sq += r[i] * r[i];
        cost += 0.5 * sq;
        if (want_J) {
            /* dense local jacobian nn x nl */
            int nl = 0;
            int* cols = (int*)malloc(sizeof(int) * (nn + 16));
            for (int b = 0; b < nb; ++b) { const int ls = pr->blk_kind[b] == GLIO_BLK_SPEEDBIAS ? 9 : 3; for (int j = 0; j < ls; ++j) cols[nl++] = goff[b] + j; }
            double* Jl = (double*)malloc(sizeof(double) * nn * nl);
            int c0 = 0;
            for (int b = 0; b < nb; ++b) {
                const int kind = pr->blk_kind[b];
                if (kind == GLIO_BLK_QUAT) {
                    double Pj[12];
                    quat_plus_jacobian(P[b], Pj);
                    for (int i = 0; i < nn; ++i)
                        for (int j = 0; j < 3; ++j) {
                            const double* row = J[b] + i * 4;
                            Jl[i * nl + c0 + j] = row[0] * Pj[j] + row[1] * Pj[3 + j] + row[2] * Pj[6 + j] + row[3] * Pj[9 + j];
                        }
                    c0 += 3;
                } else {
                    const int gs = kind == GLIO_BLK_TRANS ? 3 : 9;
                    for (int i = 0; i < nn; ++i) for (int j = 0; j < gs; ++j) Jl[i * nl + c0 + j] = J[b][i * gs + j];
                    c0 += gs;
                }
            }
            for (int a = 0; a < nl; ++a) {
                double gr = 0;
                for (int i = 0; i < nn; ++i) gr += Jl[i * nl + a] * r[i];
                if (g) g[cols[a]] += gr;
                if (H)
                    for (int bb = 0; bb < nl; ++bb) {
                        double s = 0;
                        for (int i = 0; i < nn; ++i) s += Jl[i * nl + a] * Jl[i * nl + bb];
                        H[(size_t)cols[a] * n + cols[bb]] += s;
                    }
            }
            free(Jl); free(cols);
        }
        for (int b = 0; b < nb; ++b) free(J[b]);
        free(P); free(J); free(r); free(goff);
    }

    /* 2. IMU factors (Estimator.cpp:2182-2192), no loss */
    for (int k = 0; k < p->n_imu; ++k) {
        const int i = p->imu_slot[k], j = i + 1;
        const double* P[6] = {x->trans + 3 * i, x->quat + 4 * i, x->speed_bias + 9 * i,
                              x->trans + 3 * j, x->quat + 4 * j, x->speed_bias + 9 * j};
        double r[15], J0[45], J1[60], J2[135], J3[45], J4[60], J5[135];
        double* J[6] = {J0, J1, J2, J3, J4, J5};
        if (!orc_eval_imu(&p->opts, &p->imu[k], P, r, want_J ? J : NULL)) return 0;
        rblock b = {15, 6, r, {J0, J1, J2, J3, J4, J5}, {3, 4, 9, 3, 4, 9}, {3, 3, 9, 3, 3, 9},
                    {15 * i, 15 * i + 3, 15 * i + 6, 15 * j, 15 * j + 3, 15 * j + 6},
                    {0, P[1], 0, 0, P[4], 0}};
        accumulate(&b, 0, 0, n, H, g, &cost);
    }

    /* 3. LiDAR plane factors with HuberLoss(lossKernel) (Estimator.cpp:2198-2248) */
    if (g_orc_threads > 1) {
        /* all-cores CPU baseline (bench.py): keyframes are independent (disjoint blocks of H and g); the per-keyframe
         * costs are added in keyframe order afterwards.  Not the parity path: the cost's summation order differs. */
        double* cs = (double*)calloc((size_t)W, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_orc_threads)
        for (int s = 0; s < W; ++s) lidar_keyframe(p, x, s, n, H, g, &cs[s], want_J);
        for (int s = 0; s < W; ++s) cost += cs[s];
        free(cs);
    } else {
        for (int s = 0; s < W; ++s) lidar_keyframe(p, x, s, n, H, g, &cost, want_J);
    }

    /* 4. Doppler factors with HuberLoss(1.0) (Estimator.cpp:2329-2337) */
    for (int k = 0; k < p->n_dop; ++k) {
        const glio_doppler* f = &p->dop[k];
        const int i = f->slot_i, j = f->slot_j;
        const double yaw[1] = {p->frame.yaw_enu_local};
        const double* P[7] = {x->trans + 3 * i, x->speed_bias + 9 * i, x->trans + 3 * j, x->speed_bias + 9 * j,
                              x->rcv_ddt, yaw, p->frame.anc_ecef};
        double r, J0[3], J1[9], J2[3], J3[9], J4[1];
        double* J[7] = {J0, J1, J2, J3, J4, NULL, NULL};
        orc_eval_doppler(f, P, &r, want_J ? J : NULL);
        rblock b = {1, 5, &r, {J0, J1, J2, J3, J4}, {3, 9, 3, 9, 1}, {3, 9, 3, 9, 1},
                    {15 * i, 15 * i + 6, 15 * j, 15 * j + 6, 15 * W + f->epoch}, {0, 0, 0, 0, 0}};
        accumulate(&b, 1, p->opts.doppler_huber_delta, n, H, g, &cost);
    }

    /* 5. DD pseudorange factors, no loss (Estimator.cpp:1893-1897) */
    for (int k = 0; k < p->n_dd; ++k) {
        const glio_dd_psr* f = &p->dd[k];
        const int i = f->slot_i, j = f->slot_j;
        const double yaw[1] = {p->frame.yaw_enu_local};
        const double* P[4] = {x->trans + 3 * i, x->trans + 3 * j, yaw, p->frame.anc_ecef};
        double r[19], J0[57], J1[57];
        double* J[4] = {J0, J1, NULL, NULL};
        orc_eval_dd_psr(f, P, r, want_J ? J : NULL);
        rblock b = {19, 2, r, {J0, J1}, {3, 3}, {3, 3}, {15 * i, 15 * j}, {0, 0}};
        accumulate(&b, 0, 0, n, H, g, &cost);
    }
    *cost_out = cost;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Trust-region minimiser, Ceres 1.14 TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG)
 * restated on the dense normal equations.
 */
typedef struct {
    int n;
    double radius, mu;
    int reuse;
    double alpha, dogleg_step_norm;
    double *diag, *grad, *gn;       /* D, g~ = g_s / D, Gauss-Newton step in D-space */
} dogleg;

static double vdot(const double* a, const double* b, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }

static void state_alloc(glio_state* s, int W, int n_ddt) {
    s->trans = (double*)malloc(sizeof(double) * 3 * W);
    s->quat = (double*)malloc(sizeof(double) * 4 * W);
    s->speed_bias = (double*)malloc(sizeof(double) * 9 * W);
    s->rcv_ddt = (double*)malloc(sizeof(double) * (n_ddt > 0 ? n_ddt : 1));
    s->n_ddt = n_ddt;
}
static void state_free(glio_state* s) { free(s->trans); free(s->quat); free(s->speed_bias); free(s->rcv_ddt); }
static void state_copy(glio_state* d, const glio_state* s, int W) {
    memcpy(d->trans, s->trans, sizeof(double) * 3 * W);
    memcpy(d->quat, s->quat, sizeof(double) * 4 * W);
    memcpy(d->speed_bias, s->speed_bias, sizeof(double) * 9 * W);
    if (s->n_ddt > 0) memcpy(d->rcv_ddt, s->rcv_ddt, sizeof(double) * s->n_ddt);
}
static double state_norm2(const glio_state* s, int W) {
    double a = 0;
    for (int i = 0; i < 3 * W; ++i) a += s->trans[i] * s->trans[i];
    for (int i = 0; i < 4 * W; ++i) a += s->quat[i] * s->quat[i];
    for (int i = 0; i < 9 * W; ++i) a += s->speed_bias[i] * s->speed_bias[i];
    for (int i = 0; i < s->n_ddt; ++i) a += s->rcv_ddt[i] * s->rcv_ddt[i];
    return a;
}
static double state_diff_norm2(const glio_state* a, const glio_state* b, int W, int inf) {
    double acc = 0;
#define ORC_ACC(arr, cnt) for (int i = 0; i < (cnt); ++i) { double d = a->arr[i] - b->arr[i]; if (inf) { if (fabs(d) > acc) acc = fabs(d); } else acc += d * d; }
    ORC_ACC(trans, 3 * W) ORC_ACC(quat, 4 * W) ORC_ACC(speed_bias, 9 * W) ORC_ACC(rcv_ddt, a->n_ddt)
#undef ORC_ACC
    return acc;
}

static int solve_impl(const orc_problem* p, glio_state* x, glio_summary* sum, double* history) {
    const glio_opts* o = &p->opts;
    const int W = o->window, n = 15 * W + x->n_ddt;
    const size_t nn = (size_t)n * n;
    double* H = (double*)malloc(sizeof(double) * nn);
    double* Hc = (double*)malloc(sizeof(double) * nn);
    double* Hs = (double*)malloc(sizeof(double) * nn);
    double* L = (double*)malloc(sizeof(double) * nn);
    double* g = (double*)malloc(sizeof(double) * n);
    double* gc = (double*)malloc(sizeof(double) * n);
    double* gs = (double*)malloc(sizeof(double) * n);
    double* scale = (double*)malloc(sizeof(double) * n);
    double* step = (double*)malloc(sizeof(double) * n);
    double* delta = (double*)malloc(sizeof(double) * n);
    double* tmp = (double*)malloc(sizeof(double) * n);
    double* tmp2 = (double*)malloc(sizeof(double) * n);
    dogleg dl;
    dl.n = n; dl.radius = o->initial_trust_region_radius; dl.mu = 1e-8; dl.reuse = 0; dl.alpha = 0; dl.dogleg_step_norm = 0;
    const int lm = o->trust_region_strategy == GLIO_STRATEGY_LM;
    double decrease_factor = 2.0;        /* LevenbergMarquardtStrategy::decrease_factor_ */
    dl.diag = (double*)malloc(sizeof(double) * n);
    dl.grad = (double*)malloc(sizeof(double) * n);
    dl.gn = (double*)calloc(n, sizeof(double));
    glio_state cand;
    state_alloc(&cand, W, x->n_ddt);

    memset(sum, 0, sizeof *sum);
    for (int s = 0; s < W; ++s) sum->n_lidar_residuals += p->lidar_offset[s + 1] - p->lidar_offset[s];
    double cost;
    int ok = orc_linearize(p, x, H, g, &cost);
    if (!ok) { sum->termination = GLIO_TERM_FAILURE; goto done; }
    sum->initial_cost = cost;
    /* Jacobi scaling, estimated once at iteration 0: 1/(1+sqrt(|col|^2)) */
    for (int i = 0; i < n; ++i) scale[i] = o->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[(size_t)i * n + i])) : 1.0;

    int iteration = 0, invalid = 0;
    sum->termination = GLIO_TERM_NO_CONVERGENCE;
    for (;;) {
        /* gradient_max_norm = | x - Plus(x, -g) |_inf  (EvaluateGradientAndJacobian) */
        for (int i = 0; i < n; ++i) tmp[i] = -g[i];
        orc_state_plus(x, W, tmp, &cand);
        sum->gradient_max_norm = state_diff_norm2(x, &cand, W, 1);
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (iteration >= o->max_iterations) { sum->termination = GLIO_TERM_NO_CONVERGENCE; break; }
        if (sum->gradient_max_norm <= o->gradient_tolerance) { sum->termination = GLIO_TERM_GRADIENT_TOL; break; }
        if (dl.radius <= o->min_trust_region_radius) { sum->termination = GLIO_TERM_MIN_RADIUS; break; }
        ++iteration;

        /* ---- DoglegStrategy::ComputeStep on the scaled system Hs = S H S, gs = S g */
        for (int i = 0; i < n; ++i) { gs[i] = scale[i] * g[i]; for (int j = 0; j < n; ++j) Hs[(size_t)i * n + j] = scale[i] * H[(size_t)i * n + j] * scale[j]; }
        int step_valid = 1;
        if (lm) {
            /* LevenbergMarquardtStrategy::ComputeStep (Ceres 1.14): D^2 = clamp(diag(J^T J)) / radius on the scaled
             * system, (Hs + D^2) y = gs, step = -y */
            for (int i = 0; i < n; ++i) {
                double d = Hs[(size_t)i * n + i];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                dl.diag[i] = sqrt(d);
            }
            memcpy(L, Hs, sizeof(double) * nn);
            for (int i = 0; i < n; ++i) L[(size_t)i * n + i] += dl.diag[i] * dl.diag[i] / dl.radius;
            if (chol_lower(L, n) == 0) {
                chol_solve(L, n, gs, tmp);
                for (int i = 0; i < n; ++i) { if (!isfinite(tmp[i])) step_valid = 0; step[i] = -tmp[i]; }
            } else step_valid = 0;
        } else if (!dl.reuse) {
            for (int i = 0; i < n; ++i) {
                double d = Hs[(size_t)i * n + i];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                dl.diag[i] = sqrt(d);
                dl.grad[i] = gs[i] / dl.diag[i];
            }
            /* Cauchy point: alpha = |g~|^2 / |J (g~/D)|^2 */
            for (int i = 0; i < n; ++i) tmp[i] = dl.grad[i] / dl.diag[i];
            double Jg2 = 0;
            for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += Hs[(size_t)i * n + j] * tmp[j]; Jg2 += tmp[i] * s; }
            dl.alpha = vdot(dl.grad, dl.grad, n) / Jg2;
            /* Gauss-Newton: (Hs + mu D^2) y = gs ; gn = -D y */
            int solved = 0;
            while (dl.mu < 1.0) {
                memcpy(L, Hs, sizeof(double) * nn);
                for (int i = 0; i < n; ++i) L[(size_t)i * n + i] += dl.mu * dl.diag[i] * dl.diag[i];
                if (chol_lower(L, n) == 0) {
                    chol_solve(L, n, gs, tmp);
                    int finite = 1;
                    for (int i = 0; i < n; ++i) if (!isfinite(tmp[i])) finite = 0;
                    if (finite) { solved = 1; break; }
                }
                dl.mu *= 10.0;
            }
            /* Ceres 1.14 DoglegStrategy::ComputeGaussNewtonStep: mu is only RAISED here ("next time ... the multiplier starts
             * out from the last successful solve"); it is lowered in StepAccepted alone.  When every mu < max_mu fails the
             * strategy returns LINEAR_SOLVER_FAILURE and the minimizer counts an invalid step (HandleInvalidStep). */
            if (!solved) step_valid = 0;
            else for (int i = 0; i < n; ++i) dl.gn[i] = -dl.diag[i] * tmp[i];
        }
        if (!lm) {   /* ComputeTraditionalDoglegStep */
            const double gnorm = sqrt(vdot(dl.grad, dl.grad, n));
            const double gnn = sqrt(vdot(dl.gn, dl.gn, n));
            if (gnn <= dl.radius) {
                for (int i = 0; i < n; ++i) step[i] = dl.gn[i];
                dl.dogleg_step_norm = gnn;
            } else if (gnorm * dl.alpha >= dl.radius) {
                for (int i = 0; i < n; ++i) step[i] = -(dl.radius / gnorm) * dl.grad[i];
                dl.dogleg_step_norm = dl.radius;
            } else {
                const double b_dot_a = -dl.alpha * vdot(dl.grad, dl.gn, n);
                const double a_sq = dl.alpha * dl.alpha * gnorm * gnorm;
                const double b_minus_a_sq = gnn * gnn - 2 * b_dot_a + a_sq;
                const double c = b_dot_a - a_sq;
                const double d = sqrt(c * c + b_minus_a_sq * (dl.radius * dl.radius - a_sq));
                const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (dl.radius * dl.radius - a_sq) / (d + c);
                for (int i = 0; i < n; ++i) step[i] = (-dl.alpha * (1.0 - beta)) * dl.grad[i] + beta * dl.gn[i];
                dl.dogleg_step_norm = sqrt(vdot(step, step, n));
            }
            for (int i = 0; i < n; ++i) step[i] /= dl.diag[i];
        }
        /* model_cost_change = -(gs.s + 1/2 s^T Hs s) */
        double mcc;
        {
            double lin = vdot(gs, step, n), quad = 0;
            for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += Hs[(size_t)i * n + j] * step[j]; quad += step[i] * s; }
            mcc = -(lin + 0.5 * quad);
        }
        if (!(mcc > 0.0)) step_valid = 0;
        if (!step_valid) {
            if (++invalid >= 5) { sum->termination = GLIO_TERM_FAILURE; break; }
            if (history) { double* h = history + 3 * (iteration - 1); h[0] = cost; h[1] = dl.radius; h[2] = 0.0; }
            if (lm) { dl.radius /= decrease_factor; decrease_factor *= 2.0; }
            else { dl.mu *= 10.0; dl.reuse = 0; }  /* StepIsInvalid */
            continue;
        }
        invalid = 0;
        for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
        orc_state_plus(x, W, delta, &cand);
        double ccost;
        int cok = orc_linearize(p, &cand, Hc, gc, &ccost);
        if (!cok) { if (lm) { dl.radius /= decrease_factor; decrease_factor *= 2.0; } else { dl.radius *= 0.5; dl.reuse = 1; } continue; }
        /* ParameterToleranceReached */
        {
            const double step_norm = sqrt(state_diff_norm2(x, &cand, W, 0));
            const double x_norm = sqrt(state_norm2(x, W));
            if (history) { double* h = history + 3 * (iteration - 1); h[0] = ccost; h[1] = dl.radius; h[2] = step_norm; }
            if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { sum->termination = GLIO_TERM_PARAMETER_TOL; break; }
        }
        /* FunctionToleranceReached */
        if (fabs(cost - ccost) <= o->function_tolerance * cost) { sum->termination = GLIO_TERM_FUNCTION_TOL; break; }
        const double rel = (cost - ccost) / mcc;
        if (rel > o->min_relative_decrease) {
            state_copy(x, &cand, W);
            cost = ccost;
            memcpy(H, Hc, sizeof(double) * nn);
            memcpy(g, gc, sizeof(double) * n);
            ++sum->successful_steps;
            /* StepAccepted */
            if (lm) {
                dl.radius = dl.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
                dl.radius = fmin(o->max_trust_region_radius, dl.radius);
                decrease_factor = 2.0;
            } else {
                if (rel < 0.25) dl.radius *= 0.5;
                if (rel > 0.75) dl.radius = fmax(dl.radius, 3.0 * dl.dogleg_step_norm);
                dl.mu = fmax(1e-8, 2.0 * dl.mu / 10.0);
                dl.reuse = 0;
            }
        } else if (lm) {
            dl.radius /= decrease_factor; decrease_factor *= 2.0;      /* LM StepRejected */
        } else {
            dl.radius *= 0.5; dl.reuse = 1;       /* StepRejected */
        }
    }
    sum->iterations = iteration;
    sum->final_cost = cost;
    sum->final_radius = dl.radius;
done:
    state_free(&cand);
    free(dl.diag); free(dl.grad); free(dl.gn);
    free(H); free(Hc); free(Hs); free(L); free(g); free(gc); free(gs); free(scale); free(step); free(delta); free(tmp); free(tmp2);
    return sum->termination != GLIO_TERM_FAILURE;
}

int orc_solve(const orc_problem* p, glio_state* x, glio_summary* sum) { return solve_impl(p, x, sum, NULL); }
/* the same, recording per iteration (candidate cost, radius the step was computed with, |x - candidate|): history [max_iterations][3] */
int orc_solve_history(const orc_problem* p, glio_state* x, glio_summary* sum, double* history) { return solve_impl(p, x, sum, history); }
