/*
 * orc_batch.c -- CPU ORACLE (test infrastructure): one linearisation of the batch stage's
 * scan-to-multiscan constraints (Estimator::optimizeBatchWithLandMark, GLIO/src/Estimator.cpp:
 * 3004-3076) -- BinaryLidarPlaneNormFactor (LidarKeyframeFactor.h:124-164), no loss function
 * (:2768), Ceres QuaternionParameterization on both quaternion blocks -- summed into the block-banded
 * normal equations that the multi-GPU path all-reduces.  PARITY UNPINNED -- see glio_oracle.h.
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
