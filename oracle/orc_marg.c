/*
 * orc_marg.c -- CPU ORACLE (test infrastructure): the marginalization step that follows every
 * sliding-window solve (GLIO/src/Estimator.cpp:2462-2607) through MarginalizationInfo::
 * {AddResidualBlockInfo, PreMarginalize, Marginalize, GetParameterBlocks}
 * (GLIO/src/MarginalizationFactor.cpp:87-221) and ResidualBlockInfo::Evaluate (:31-71).
 *
 * Factors added by the reference: the previous prior (drop = its slot-0 blocks, :2464-2480), the
 * IMU factor between slots 0 and 1 (drop T0,Q0,SB0, :2521-2534) and the LiDAR plane factors of ALL
 * W frames (only frame 0's carry a drop set, quirk Q7, :2538-2568).  Quaternion blocks use the
 * "global Jacobian minus the w column" convention (quirk Q8, MarginalizationFactor.cpp:9-17).
 *
 * The reference orders blocks by unordered_map iteration (address keyed, unspecified); this
 * restatement fixes: dropped = [T0 Q0 SB0] (m = 15), kept = [T1 Q1 SB1 T2 Q2 ... T(W-1) Q(W-1)]
 * (n = 6(W-1)+9).  J0^T J0 and J0^T r0 are invariant to that choice and to eigenvector signs; compare
 * those, not J0 itself.  Pinned on the reference's own MarginalizationInfo (oracle/_ref, tests/test_oracle_ref.py) through these invariants; see glio_oracle.h.
 */
#include <stdlib.h>
#include "glio_oracle.h"
#include "orc_math.h"

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix: A = V diag(w) V^T, V row-major with
 * eigenvectors in columns (stands in for Eigen::SelfAdjointEigenSolver) */
static void sym_eig(double* A, int n, double* w, double* V) {
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

/* index of (slot, kind) in the fixed ordering; returns local offset */
static int marg_off(int slot, int kind) {
    if (slot == 0) return kind == GLIO_BLK_TRANS ? 0 : (kind == GLIO_BLK_QUAT ? 3 : 6);
    if (slot == 1) return 15 + (kind == GLIO_BLK_TRANS ? 0 : (kind == GLIO_BLK_QUAT ? 3 : 6));
    return 30 + 6 * (slot - 2) + (kind == GLIO_BLK_TRANS ? 0 : 3);
}

static void add_block(double* A, double* b, int pos, int nr, const double* r, int nb, double* const* Jloc,
                      const int* lsz, const int* off) {
    /* ThreadsConstructA (MarginalizationFactor.cpp:3-29) */
    for (int i = 0; i < nb; ++i) {
        for (int j = i; j < nb; ++j)
            for (int a = 0; a < lsz[i]; ++a)
                for (int c = 0; c < lsz[j]; ++c) {
                    double s = 0;
                    for (int k = 0; k < nr; ++k) s += Jloc[i][k * lsz[i] + a] * Jloc[j][k * lsz[j] + c];
                    A[(size_t)(off[i] + a) * pos + off[j] + c] += s;
                    if (i != j) A[(size_t)(off[j] + c) * pos + off[i] + a] += s;
                }
        for (int a = 0; a < lsz[i]; ++a) {
            double s = 0;
            for (int k = 0; k < nr; ++k) s += Jloc[i][k * lsz[i] + a] * r[k];
            b[off[i] + a] += s;
        }
    }
}

int orc_marginalize(const orc_problem* p, const glio_state* x, double* lin_jac, double* lin_res,
                    int32_t* blk_slot, int32_t* blk_kind, int32_t* blk_idx, double* blk_x0) {
    const int W = p->opts.window;
    const int m = 15, n = 6 * (W - 1) + 9, pos = m + n;
    const double eps = 1e-8;                                   /* MarginalizationFactor.h: eps */
    double* A = (double*)calloc((size_t)pos * pos, sizeof(double));
    double* b = (double*)calloc(pos, sizeof(double));

    /* (1) previous prior */
    if (p->prior.n > 0) {
        const glio_prior* pr = &p->prior;
        const int nn = pr->n, nb = pr->n_blocks;
        const double** P = (const double**)malloc(sizeof(double*) * nb);
        double** J = (double**)malloc(sizeof(double*) * nb);
        double** Jl = (double**)malloc(sizeof(double*) * nb);
        int* lsz = (int*)malloc(sizeof(int) * nb);
        int* off = (int*)malloc(sizeof(int) * nb);
        double* r = (double*)malloc(sizeof(double) * nn);
        for (int k = 0; k < nb; ++k) {
            const int s = pr->blk_slot[k], kind = pr->blk_kind[k];
            const int gs = kind == GLIO_BLK_TRANS ? 3 : (kind == GLIO_BLK_QUAT ? 4 : 9);
            P[k] = kind == GLIO_BLK_TRANS ? x->trans + 3 * s : (kind == GLIO_BLK_QUAT ? x->quat + 4 * s : x->speed_bias + 9 * s);
            J[k] = (double*)malloc(sizeof(double) * nn * gs);
            lsz[k] = gs == 4 ? 3 : gs;
            off[k] = marg_off(s, kind);
            Jl[k] = (double*)malloc(sizeof(double) * nn * lsz[k]);
        }
        orc_eval_marg(pr, P, r, J);
        for (int k = 0; k < nb; ++k) {
            const int gs = pr->blk_kind[k] == GLIO_BLK_QUAT ? 4 : lsz[k];
            for (int i = 0; i < nn; ++i) for (int c = 0; c < lsz[k]; ++c) Jl[k][i * lsz[k] + c] = J[k][i * gs + (gs - lsz[k]) + c];  /* rightCols */
        }
        add_block(A, b, pos, nn, r, nb, Jl, lsz, off);
        for (int k = 0; k < nb; ++k) { free(J[k]); free(Jl[k]); }
        free(P); free(J); free(Jl); free(lsz); free(off); free(r);
    }
    /* (2) IMU factor (0,1) */
    for (int k = 0; k < p->n_imu; ++k) {
        if (p->imu_slot[k] != 0) continue;
        const double* P[6] = {x->trans, x->quat, x->speed_bias, x->trans + 3, x->quat + 4, x->speed_bias + 9};
        double r[15], J0[45], J1[60], J2[135], J3[45], J4[60], J5[135];
        double* J[6] = {J0, J1, J2, J3, J4, J5};
        orc_eval_imu(&p->opts, &p->imu[k], P, r, J);
        double Q1[45], Q4[45];
        for (int i = 0; i < 15; ++i) for (int c = 0; c < 3; ++c) { Q1[i * 3 + c] = J1[i * 4 + 1 + c]; Q4[i * 3 + c] = J4[i * 4 + 1 + c]; }
        double* Jl[6] = {J0, Q1, J2, J3, Q4, J5};
        const int lsz[6] = {3, 3, 9, 3, 3, 9};
        const int off[6] = {marg_off(0, 0), marg_off(0, 1), marg_off(0, 2), marg_off(1, 0), marg_off(1, 1), marg_off(1, 2)};
        add_block(A, b, pos, 15, r, 6, Jl, lsz, off);
    }
    /* (3) LiDAR factors of all frames with Huber (ResidualBlockInfo::Evaluate :44-70) */
    for (int s = 0; s < W; ++s) {
        const double* P[2] = {x->trans + 3 * s, x->quat + 4 * s};
        const int ot = marg_off(s, GLIO_BLK_TRANS), oq = marg_off(s, GLIO_BLK_QUAT);
        for (int i = p->lidar_offset[s]; i < p->lidar_offset[s + 1]; ++i) {
            double r, Jt[3], Jq[4];
            double* J[2] = {Jt, Jq};
            orc_eval_lidar_plane(&p->opts, p->lidar_pts + 4 * (size_t)i, p->lidar_planes + 4 * (size_t)i, p->lidar_scores[i], P, &r, J);
            const double sq = r * r, a = p->opts.huber_delta;
            double rho1 = 1.0;
            if (sq > a * a) rho1 = a / sqrt(sq);
            const double sr = sqrt(rho1);           /* rho'' <= 0 branch: scaling only */
            double Jl[6] = {sr * Jt[0], sr * Jt[1], sr * Jt[2], sr * Jq[1], sr * Jq[2], sr * Jq[3]};
            const double rc = sr * r;
            const int idx[6] = {ot, ot + 1, ot + 2, oq, oq + 1, oq + 2};
            for (int u = 0; u < 6; ++u) {
                b[idx[u]] += Jl[u] * rc;
                for (int v = 0; v < 6; ++v) A[(size_t)idx[u] * pos + idx[v]] += Jl[u] * Jl[v];
            }
        }
    }
    /* Marginalize (:176-201) */
    double* Amm = (double*)malloc(sizeof(double) * m * m);
    double wm[15], Vm[225], Amm_inv[225];
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm[i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
    sym_eig(Amm, m, wm, Vm);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0;
            for (int k = 0; k < m; ++k) s += Vm[i * m + k] * (wm[k] > eps ? 1.0 / wm[k] : 0.0) * Vm[j * m + k];
            Amm_inv[i * m + j] = s;
        }
    /* A_rr - A_rm Amm^-1 A_mr ; b_rr - A_rm Amm^-1 b_mm */
    double* T = (double*)malloc(sizeof(double) * n * m);      /* A_rm * Amm_inv */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0;
            for (int k = 0; k < m; ++k) s += A[(size_t)(m + i) * pos + k] * Amm_inv[k * m + j];
            T[i * m + j] = s;
        }
    double* S = (double*)malloc(sizeof(double) * n * n);
    double* bs = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < m; ++k) s += T[i * m + k] * A[(size_t)k * pos + m + j];
            S[i * n + j] = A[(size_t)(m + i) * pos + m + j] - s;
        }
        double s = 0;
        for (int k = 0; k < m; ++k) s += T[i * m + k] * b[k];
        bs[i] = b[m + i] - s;
    }
    double* w2 = (double*)malloc(sizeof(double) * n);
    double* V2 = (double*)malloc(sizeof(double) * n * n);
    double* Scopy = (double*)malloc(sizeof(double) * n * n);
    /* SelfAdjointEigenSolver reads the lower triangle */
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Scopy[i * n + j] = (j <= i) ? S[i * n + j] : S[j * n + i];
    sym_eig(Scopy, n, w2, V2);
    for (int k = 0; k < n; ++k) {
        const double sv = w2[k] > eps ? w2[k] : 0.0;
        const double sinv = w2[k] > eps ? 1.0 / w2[k] : 0.0;
        const double ss = sqrt(sv), sis = sqrt(sinv);
        double acc = 0;
        for (int j = 0; j < n; ++j) { lin_jac[(size_t)k * n + j] = ss * V2[(size_t)j * n + k]; acc += V2[(size_t)j * n + k] * bs[j]; }
        lin_res[k] = sis * acc;
    }
    /* GetParameterBlocks with addr_shift i -> i-1 (Estimator.cpp:2584-2600) */
    int nb = 0;
    for (int s = 1; s < W; ++s) {
        const int kinds = (s == 1) ? 3 : 2;
        for (int k = 0; k < kinds; ++k) {
            blk_slot[nb] = s - 1;
            blk_kind[nb] = k;
            blk_idx[nb] = marg_off(s, k) - m;
            const double* src = k == 0 ? x->trans + 3 * s : (k == 1 ? x->quat + 4 * s : x->speed_bias + 9 * s);
            const int gs = k == 0 ? 3 : (k == 1 ? 4 : 9);
            for (int c = 0; c < 9; ++c) blk_x0[9 * nb + c] = c < gs ? src[c] : 0.0;
            ++nb;
        }
    }
    free(A); free(b); free(Amm); free(T); free(S); free(bs); free(w2); free(V2); free(Scopy);
    return n;
}
