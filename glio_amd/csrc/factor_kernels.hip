// factor_kernels.hip -- K4/K5/K6 + assembly: the O(W) "small" factors of the sliding-window problem
// and the gather that builds the dense normal equations H, g on device.
//
//   k_small_factors : workgroup roles (inside glio_solve they are the first workgroups of k_linearize_all, which runs
//                     the K3 workgroups beside them; the roles overlay one LDS pool)
//       [0, n_imu)          ImuFactor::Evaluate          (reference GLIO/include/factors/ImuFactor.h:21-171,
//                                                          Preintegration.h:196-235) -> 30x30 block
//       [.., +n_groups)     dd_psr_factor_20::Evaluate   (dd_psr_factor.hpp:25-171) and
//                           tcdopplerFactor              (dopp_factor.hpp:24-75, HuberLoss(1.0)) of one
//                           (slot_i,slot_j) pair -> 30x30 block + per-epoch clock-drift coupling
//       last 1+24           MarginalizationFactor::Evaluate (GLIO/src/MarginalizationFactor.cpp:233-287):
//                           one workgroup for r, g, cost and 24 sharing the rows of H
//   k_assemble      : H[r][c] / g[r] gathered from those blocks and from the K3 partials (no atomics, deterministic)
//
// All Jacobians go through the same chain as Ceres: global Jacobian -> (loss corrector) ->
// QuaternionParameterization Jacobian (left (+), GraphGNSSLibV1.1/docs/source/nnls_modeling.rst:1312-1327).
// These factors are tiny (LDS/latency-bound, not roofline material); the point of having them on device
// is that a whole trust-region solve runs without a host round trip.
#include "k3_device.h"
#include "batch_device.h"

#define SF_THREADS 256

struct SmallArgs {
    long long* dbg;           // per-workgroup duration (100 MHz ticks), development aid
    int W, n_imu, n_groups, has_prior, n_ddt;
    int marg;                  // 1 = marginalization convention for quaternion blocks (global x,y,z columns, quirk Q8)
    const double* x0; const double* x1;
    const SolverStatus* st; int use_status; int fixed_which;
    const ImuEdgeDev* imu; PairBlock* imu_blocks;           // [2][W]
    const glio_dd_psr* dd; const glio_doppler* dop; const GnssGroup* groups; const DopRun* runs; int n_runs;
    PairBlock* gnss_blocks; DdtBlock* ddt_blocks; int gnss_stride; int ddt_stride;
    double gravity, dop_huber;
    double R_ecef_local[9]; double anc[3];
    // prior
    int np, npb;
    const double* pJ0; const double* pA0; const double* pr0; const double* px0;
    const int* pslot; const int* pkind; const int* pidx; const int* pcolblk;
    double* pH; double* pg; double* pcost; double* pwork;
    // chain-layout contributions (glio_device.h, GLIO_CS_*): written beside the pair blocks, consumed by k_chain_step
    double* chain_src; const short* chain_tabs; int pair_H;
};
__device__ __forceinline__ double* chain_slice(const SmallArgs& a, const int which, const int slot, const int source) {
    return a.chain_src + (((size_t)which * a.W + slot) * GLIO_CS_SOURCES + source) * GLIO_CS_STRIDE;
}

// ------------------------------------------------------------------------------------------------
// IMU
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void qleft16(const double q[4], double M[16]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    M[0] = w; M[1] = -x; M[2] = -y; M[3] = -z;
    M[4] = x; M[5] = w; M[6] = -z; M[7] = y;
    M[8] = y; M[9] = z; M[10] = w; M[11] = -x;
    M[12] = z; M[13] = -y; M[14] = x; M[15] = w;
}
__device__ __forceinline__ void qright16(const double p[4], double M[16]) {
    const double w = p[0], x = p[1], y = p[2], z = p[3];
    M[0] = w; M[1] = -x; M[2] = -y; M[3] = -z;
    M[4] = x; M[5] = w; M[6] = z; M[7] = -y;
    M[8] = y; M[9] = -z; M[10] = w; M[11] = x;
    M[12] = z; M[13] = y; M[14] = -x; M[15] = w;
}

// global column offsets of the six parameter blocks Pi3 Qi4 SBi9 Pj3 Qj4 SBj9
#define IMU_GC 32
struct ImuLds { double Jg[15 * IMU_GC], Jl[15 * 30], WJ[15 * 30], S[225], r[15], wr[15]; };
// `eval_out` != NULL: single-factor evaluator mode (glio_eval_imu): write the whitened residual [15] and
// the whitened GLOBAL Jacobians [15][32] (Pi3 Qi4 SBi9 Pj3 Qj4 SBj9) and return.
__device__ __forceinline__ void imu_block(const double gravity, const double* __restrict__ pPi, const double* __restrict__ pQi,
                          const double* __restrict__ pSBi, const double* __restrict__ pPj, const double* __restrict__ pQj,
                          const double* __restrict__ pSBj, const ImuEdgeDev& e, PairBlock* out, double* eval_out, const int marg, unsigned char* pool,
                          double* cs_a = nullptr, double* cs_b = nullptr, long long* stamp_dbg = nullptr, const bool pair_H = true) {
#ifdef GLIO_DEV_STAMPS
#define IMU_STAMP(k) do { if (stamp_dbg && threadIdx.x == 0) stamp_dbg[k] = wall_clock64(); } while (0)
#else
#define IMU_STAMP(k) do { } while (0)
#endif
    IMU_STAMP(0);
    // LDS comes from the caller's pool: the roles of the small-factor kernel overlay one another (a workgroup has one role)
    ImuLds& lds_ = *reinterpret_cast<ImuLds*>(pool);
    double (&Jg)[15 * IMU_GC] = lds_.Jg; double (&Jl)[15 * 30] = lds_.Jl; double (&WJ)[15 * 30] = lds_.WJ;
    double (&S)[225] = lds_.S; double (&r)[15] = lds_.r; double (&wr)[15] = lds_.wr;
    const int tid = threadIdx.x;
    const int i = e.slot_i, j = i + 1;
    enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };

    // ---- common quantities, computed redundantly by every lane (uniform control flow)
    double Pi[3], Pj[3], Qi_raw[4], Qj_raw[4], Qi[4], Qj[4], Vi[3], Vj[3], Bai[3], Bgi[3], Baj[3], Bgj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Pi[k] = pPi[k]; Pj[k] = pPj[k];
        Vi[k] = pSBi[k]; Bai[k] = pSBi[3 + k]; Bgi[k] = pSBi[6 + k];
        Vj[k] = pSBj[k]; Baj[k] = pSBj[3 + k]; Bgj[k] = pSBj[6 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { Qi_raw[k] = Qi[k] = pQi[k]; Qj_raw[k] = Qj[k] = pQj[k]; }
    d_qnormalize(Qi); d_qnormalize(Qj);                       // ImuFactor.h:25,33
    const double g[3] = {0.0, 0.0, -gravity};
    const double dt = e.sum_dt;
    double dba[3], dbg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { dba[k] = Bai[k] - e.lin_ba[k]; dbg[k] = Bgi[k] - e.lin_bg[k]; }
    double th[3], dq[4], cdq[4], cdv[3], cdp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        th[k] = e.dq_dbg[3 * k] * dbg[0] + e.dq_dbg[3 * k + 1] * dbg[1] + e.dq_dbg[3 * k + 2] * dbg[2];
        cdv[k] = e.delta_v[k] + (e.dv_dba[3 * k] * dba[0] + e.dv_dba[3 * k + 1] * dba[1] + e.dv_dba[3 * k + 2] * dba[2])
                 + (e.dv_dbg[3 * k] * dbg[0] + e.dv_dbg[3 * k + 1] * dbg[1] + e.dv_dbg[3 * k + 2] * dbg[2]);
        cdp[k] = e.delta_p[k] + (e.dp_dba[3 * k] * dba[0] + e.dp_dba[3 * k + 1] * dba[1] + e.dp_dba[3 * k + 2] * dba[2])
                 + (e.dp_dbg[3 * k] * dbg[0] + e.dp_dbg[3 * k + 1] * dbg[1] + e.dp_dbg[3 * k + 2] * dbg[2]);
    }
    dq[0] = 1.0; dq[1] = th[0] / 2.0; dq[2] = th[1] / 2.0; dq[3] = th[2] / 2.0;   // deltaQ, not normalised (Q4)
    d_qmul(e.delta_q, dq, cdq);
    double Qi_inv[4], Qj_inv[4], cdq_inv[4], tmp[3], tmp1[3];
    d_qinv(Qi, Qi_inv); d_qinv(Qj, Qj_inv); d_qinv(cdq, cdq_inv);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tmp[k] = -0.5 * g[k] * dt * dt + Pj[k] - Pi[k] - Vi[k] * dt;
        tmp1[k] = -g[k] * dt + Vj[k] - Vi[k];
    }
    double Ri_inv[9];
    d_q2R(Qi_inv, Ri_inv);

    // ---- residual (Preintegration.h:227-232) by lane 0; zero Jg by everyone
    for (int k = tid; k < 15 * IMU_GC; k += SF_THREADS) Jg[k] = 0.0;
    for (int k = tid; k < 225; k += SF_THREADS) S[k] = e.sqrt_info[k];
    if (tid == 0) {
        double rot[3], qij[4], qe[4];
        d_qrot(Qi_inv, tmp, rot);
        for (int k = 0; k < 3; ++k) r[O_P + k] = rot[k] - cdp[k];
        d_qmul(Qi_inv, Qj, qij);
        d_qmul(cdq_inv, qij, qe);
        d_qnormalize(qe);
        for (int k = 0; k < 3; ++k) r[O_R + k] = 2.0 * qe[1 + k];
        d_qrot(Qi_inv, tmp1, rot);
        for (int k = 0; k < 3; ++k) r[O_V + k] = rot[k] - cdv[k];
        for (int k = 0; k < 3; ++k) { r[O_BA + k] = Baj[k] - Bai[k]; r[O_BG + k] = Bgj[k] - Bgi[k]; }
    }
    __syncthreads();
    IMU_STAMP(1);

    // ---- global Jacobians, one parameter block per role (ImuFactor.h:63-167).  The roles run different code, so they are
    //      spread over the four wavefronts (lane 0 of each): inside one wavefront they would execute one after the other.
    const int role_wave = tid >> 6;
    const bool role_lane = (tid & 63) == 0;
    if (role_lane && role_wave == 0) {            // Pi
        for (int p = 0; p < 3; ++p) for (int c = 0; c < 3; ++c) Jg[(O_P + p) * IMU_GC + 0 + c] = -Ri_inv[p * 3 + c];
    }
    if (role_lane && role_wave == 1) {     // Qi  (the P,V rows are the reference's as-written block, quirk Q15)
        const double w = Qi[0];
        const double* u = Qi + 1;
        for (int sblk = 0; sblk < 2; ++sblk) {
            const double* v = sblk == 0 ? tmp : tmp1;
            const int row = sblk == 0 ? O_P : O_V;
            double uxv[3];
            d_cross(u, v, uxv);
            const double udv = d_dot3(u, v);
            const double Sv[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
            for (int p = 0; p < 3; ++p) {
                Jg[(row + p) * IMU_GC + 3] = 2 * (w * v[p] + uxv[p]);
                for (int c = 0; c < 3; ++c)
                    Jg[(row + p) * IMU_GC + 4 + c] = 2 * ((p == c ? udv : 0.0) + u[p] * v[c] - v[p] * u[c] - w * Sv[p * 3 + c]);
            }
        }
        double L[16], Rm[16];
        qleft16(Qj_inv, L); qright16(cdq, Rm);
        for (int p = 0; p < 3; ++p)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += L[(1 + p) * 4 + k] * Rm[k * 4 + c];
                Jg[(O_R + p) * IMU_GC + 3 + c] = -2 * s;
            }
    }
    if (role_lane && role_wave == 2) {     // SBi
        double qa[4], qb[4];
        d_qmul(Qj_inv, Qi, qa);
        d_qmul(qa, cdq, qb);
        const double TL[9] = {qb[0], -qb[3], qb[2], qb[3], qb[0], -qb[1], -qb[2], qb[1], qb[0]};   // w I + [vec]x
        for (int p = 0; p < 3; ++p)
            for (int c = 0; c < 3; ++c) {
                Jg[(O_P + p) * IMU_GC + 7 + 0 + c] = -Ri_inv[p * 3 + c] * dt;
                Jg[(O_P + p) * IMU_GC + 7 + 3 + c] = -e.dp_dba[p * 3 + c];
                Jg[(O_P + p) * IMU_GC + 7 + 6 + c] = -e.dp_dbg[p * 3 + c];
                Jg[(O_V + p) * IMU_GC + 7 + 0 + c] = -Ri_inv[p * 3 + c];
                Jg[(O_V + p) * IMU_GC + 7 + 3 + c] = -e.dv_dba[p * 3 + c];
                Jg[(O_V + p) * IMU_GC + 7 + 6 + c] = -e.dv_dbg[p * 3 + c];
                double s = 0;
                for (int k = 0; k < 3; ++k) s += TL[p * 3 + k] * e.dq_dbg[k * 3 + c];
                Jg[(O_R + p) * IMU_GC + 7 + 6 + c] = -s;
            }
        for (int p = 0; p < 3; ++p) { Jg[(O_BA + p) * IMU_GC + 7 + 3 + p] = -1.0; Jg[(O_BG + p) * IMU_GC + 7 + 6 + p] = -1.0; }
    }
    if (role_lane && role_wave == 0) {     // Pj
        for (int p = 0; p < 3; ++p) for (int c = 0; c < 3; ++c) Jg[(O_P + p) * IMU_GC + 16 + c] = Ri_inv[p * 3 + c];
    }
    if (role_lane && role_wave == 3) {     // Qj
        double qa[4], L[16];
        d_qmul(cdq_inv, Qi_inv, qa);
        qleft16(qa, L);
        for (int p = 0; p < 3; ++p) for (int c = 0; c < 4; ++c) Jg[(O_R + p) * IMU_GC + 19 + c] = 2 * L[(1 + p) * 4 + c];
    }
    if (role_lane && role_wave == 0) {     // SBj
        for (int p = 0; p < 3; ++p) for (int c = 0; c < 3; ++c) Jg[(O_V + p) * IMU_GC + 23 + c] = Ri_inv[p * 3 + c];
        for (int p = 0; p < 3; ++p) { Jg[(O_BA + p) * IMU_GC + 23 + 3 + p] = 1.0; Jg[(O_BG + p) * IMU_GC + 23 + 6 + p] = 1.0; }
    }
    __syncthreads();
    IMU_STAMP(2);

    if (eval_out) {
        for (int idx = tid; idx < 15 * IMU_GC; idx += SF_THREADS) {
            const int rr = idx / IMU_GC, c = idx % IMU_GC;
            double sacc = 0;
            for (int k = 0; k < 15; ++k) sacc += S[rr * 15 + k] * Jg[k * IMU_GC + c];
            eval_out[15 + idx] = sacc;
        }
        if (tid < 15) {
            double sacc = 0;
            for (int k = 0; k < 15; ++k) sacc += S[tid * 15 + k] * r[k];
            eval_out[tid] = sacc;
        }
        return;
    }
    // ---- local parameterisation (Plus Jacobian at the RAW quaternion, as Ceres does)
    double PJi[12], PJj[12];
    d_plus_jac(Qi_raw, PJi); d_plus_jac(Qj_raw, PJj);
    for (int idx = tid; idx < 450; idx += SF_THREADS) {
        const int rr = idx / 30, c = idx % 30;
        const double* row = Jg + rr * IMU_GC;
        double v;
        if (c < 3) v = row[c];
        else if (c < 6) { const int cc = c - 3; v = marg ? row[4 + cc] : row[3] * PJi[cc] + row[4] * PJi[3 + cc] + row[5] * PJi[6 + cc] + row[6] * PJi[9 + cc]; }
        else if (c < 15) v = row[7 + (c - 6)];
        else if (c < 18) v = row[16 + (c - 15)];
        else if (c < 21) { const int cc = c - 18; v = marg ? row[20 + cc] : row[19] * PJj[cc] + row[20] * PJj[3 + cc] + row[21] * PJj[6 + cc] + row[22] * PJj[9 + cc]; }
        else v = row[23 + (c - 21)];
        Jl[idx] = v;
    }
    __syncthreads();
    IMU_STAMP(3);
    // ---- whitening by sqrt_info (ImuFactor.h:47,69,97,...)
    for (int idx = tid; idx < 450; idx += SF_THREADS) {
        const int rr = idx / 30, c = idx % 30;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s += S[rr * 15 + k] * Jl[k * 30 + c];
        WJ[idx] = s;
    }
    if (tid < 15) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += S[tid * 15 + k] * r[k];
        wr[tid] = s;
    }
    __syncthreads();
    IMU_STAMP(4);
    for (int idx = tid; idx < 900; idx += SF_THREADS) {
        const int p = idx / 30, c = idx % 30;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s += WJ[k * 30 + p] * WJ[k * 30 + c];
        if (pair_H) out->H[idx] = s;
        if (cs_a) {          // the same entry in chain layout: aa -> D of slot_i, ba -> B of slot_i, bb -> D of slot_j
            if (p < 15) { if (c <= p) cs_a[p * (p + 1) / 2 + c] = s; }
            else if (c < 15) cs_a[120 + (p - 15) * 15 + c] = s;
            else if (c <= p) cs_b[(p - 15) * (p - 14) / 2 + (c - 15)] = s;
        }
    }
    if (tid < 30) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += WJ[k * 30 + tid] * wr[k];
        out->g[tid] = s;
    }
    if (tid == 32) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += wr[k] * wr[k];
        out->cost = 0.5 * s;
        out->slot_a = i; out->slot_b = j;
    }
    IMU_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// GNSS group: DD pseudorange (no loss) + Doppler (Huber) of one (slot_i, slot_j) pair
// ------------------------------------------------------------------------------------------------
#define DOP_CHUNK 128
#define DD_CHUNK 8           /* DD factors evaluated side by side: 32 lanes each */
#define GN_MAX_RUNS 32       /* Doppler epochs of one keyframe pair kept in LDS (more: read from global) */
struct GnssLds {
    double raw[DD_CHUNK][19], Jri[DD_CHUNK][19 * 3], Jrj[DD_CHUNK][19 * 3];
    double wE[DD_CHUNK][19 * 8];          // whitened rows: 6 Jacobian entries, the residual, (pad)
    double sWt[DD_CHUNK][19 * 19];
    double sE[DD_CHUNK][20][3], sRu[DD_CHUNK][20], sRr[DD_CHUNK][20], sObs[DD_CHUNK][20];   // per satellite: e^T R, |d_u|, |d_r|, psr_u - psr_r
    double dE[DOP_CHUNK * 16];            // per row: 13 Jacobian entries, corrected residual, rho, 1
    double s_cost[2];
    double sH6[36];                       // the DD part of the pair block, for the chain-layout slices
    double dpart[DD_CHUNK][44];           // per-factor parts of the DD reduction
    double sG6[8];                        // DD gradient part (6), met with the Doppler part in the scatter
    double sU[92];                        // Doppler sums by UNIQUE entry (78 of the symmetric 12 x 12, 12 of g, the cost), met with their readers in the scatter
    DopRun s_runs[GN_MAX_RUNS];
    int s_nw[DD_CHUNK], s_m[DD_CHUNK];
};
__device__ void gnss_block(const SmallArgs& a, const double* __restrict__ x, const GnssGroup& gr, int gidx,
                           PairBlock* out, DdtBlock* ddt_out, unsigned char* pool, const int which) {
    // rounds like the reference's scalar build: no FMA contraction anywhere in the GNSS role (double differences of ~2.6e7 m ranges; O(10^2) rows,
    // no measurable cost) -- tests/test_golden_ref.py holds its residuals to the reference's vectors at 1e-11
#pragma clang fp contract(off)
    // All factors of the pair are evaluated SIDE BY SIDE (DD factor f -> lanes 32 f' .. 32 f' + 31, Doppler row -> one
    // lane).  These are chains of dependent global loads (~1-2 us each on this part), so what counts is the number of
    // latency ROUNDS, not the arithmetic: everything a factor needs is fetched in one round (per-satellite quantities by
    // the satellite's own lane, the master's reach the others through LDS; the whitening matrix cooperatively into LDS;
    // the clock-drift unknowns and the epoch table of the pair too).  Sums keep the per-factor association of the
    // sequential formulation.
    GnssLds& lds_ = *reinterpret_cast<GnssLds*>(pool);
    double (&raw)[DD_CHUNK][19] = lds_.raw; double (&Jri)[DD_CHUNK][19 * 3] = lds_.Jri; double (&Jrj)[DD_CHUNK][19 * 3] = lds_.Jrj;
    double (&wE)[DD_CHUNK][19 * 8] = lds_.wE; double (&sWt)[DD_CHUNK][19 * 19] = lds_.sWt;
    double (&sE)[DD_CHUNK][20][3] = lds_.sE; double (&sRu)[DD_CHUNK][20] = lds_.sRu; double (&sRr)[DD_CHUNK][20] = lds_.sRr; double (&sObs)[DD_CHUNK][20] = lds_.sObs;
    int (&s_nw)[DD_CHUNK] = lds_.s_nw; int (&s_m)[DD_CHUNK] = lds_.s_m;
    double (&dE)[DOP_CHUNK * 16] = lds_.dE; DopRun (&s_runs)[GN_MAX_RUNS] = lds_.s_runs;
    double (&s_cost)[2] = lds_.s_cost;
    const int tid = threadIdx.x;
    const int W = a.W;
    const int si = gr.slot_i, sj = gr.slot_j;
    double Pi[3], Pj[3], Vi[3], Vj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Pi[k] = x[3 * si + k]; Pj[k] = x[3 * sj + k];
        Vi[k] = x[7 * W + 9 * si + k]; Vj[k] = x[7 * W + 9 * sj + k];
    }
#ifdef GLIO_DEV_STAMPS
#define GN_STAMP(k) do { if (gidx == 0 && tid == 0 && a.dbg) a.dbg[200 + (k)] = wall_clock64(); } while (0)
#else
#define GN_STAMP(k) do { } while (0)
#endif
    GN_STAMP(0);
    const int n_runs = gr.run_end - gr.run_begin;
    if (tid < n_runs && tid < GN_MAX_RUNS) s_runs[tid] = a.runs[gr.run_begin + tid];
    const double* R = a.R_ecef_local;
    // thread-private accumulators (thread p owns one entry), summed over factors in fixed order
    double h6 = 0.0, g6 = 0.0, cost_dd = 0.0;       // DD: p<36 -> H6[p]; 36<=p<42 -> g6; p==42 cost
    double dq = 0.0;                                // Doppler: lane 2 e owns unique entry e (78 of H12's upper triangle row by row, 12 of g12, the cost)

    // The Doppler rows of the first chunk are asked for NOW, in the round trip of the DD factors' data: they depend on nothing computed here, and asked for
    // behind the DD part they were one more round trip (~1.5 us) of the longest role of the launch.  (Scalars only: a copy of the record itself lived in scratch.)
    struct DopRow { double ratio, var, p0, p1, p2, v0, v1, v2, sv_ddt, doppler, lamda, l0, l1, l2, R0, R1, R2, R3, R4, R5, R6, R7, R8; int epoch; };
    auto load_row = [&](const glio_doppler& F) {
        DopRow r;
        r.ratio = F.ratio; r.var = F.var; r.p0 = F.sat_pos[0]; r.p1 = F.sat_pos[1]; r.p2 = F.sat_pos[2]; r.v0 = F.sat_vel[0]; r.v1 = F.sat_vel[1]; r.v2 = F.sat_vel[2];
        r.sv_ddt = F.sv_ddt; r.doppler = F.doppler; r.lamda = F.lamda; r.l0 = F.lever_arm[0]; r.l1 = F.lever_arm[1]; r.l2 = F.lever_arm[2];
        r.R0 = F.R_ecef_local[0]; r.R1 = F.R_ecef_local[1]; r.R2 = F.R_ecef_local[2]; r.R3 = F.R_ecef_local[3]; r.R4 = F.R_ecef_local[4];
        r.R5 = F.R_ecef_local[5]; r.R6 = F.R_ecef_local[6]; r.R7 = F.R_ecef_local[7]; r.R8 = F.R_ecef_local[8]; r.epoch = F.epoch;
        return r;
    };
    const int dop_cnt0 = min(DOP_CHUNK, gr.dop_end - gr.dop_begin);
    DopRow rowp = {};
    if (tid < dop_cnt0) rowp = load_row(a.dop[gr.dop_begin + tid]);
    double ddt_p = 0.0;          // x[16 W + epoch of the row]: needs the row; asked for behind the DD part's first stage
    bool ddt_have = false;
    // ---- DD pseudorange factors (dd_psr_factor.hpp:25-171), DD_CHUNK at a time
    for (int f0 = gr.dd_begin; f0 < gr.dd_end; f0 += DD_CHUNK) {
        const int nf = min(DD_CHUNK, gr.dd_end - f0);
        const int fl = tid >> 5, i = tid & 31;
        double f_ratio = 0.0, f_thr = 0.0;
        if (fl < nf) {
            const glio_dd_psr& F = a.dd[f0 + fl];
            f_ratio = F.ratio; f_thr = F.threshold;
            // the whitening matrix, cooperatively (independent of n_sat: the used part is the leading nw x nw block)
            for (int k = i; k < 19 * 19; k += 32) sWt[fl][k] = F.weight[k];
            if (i == 0) { s_nw[fl] = F.n_sat - 1; s_m[fl] = F.master; }
            if (i < GLIO_DD_MAX_SAT) {
                double Pe[3], lp[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) lp[k] = f_ratio * Pi[k] + (1.0 - f_ratio) * Pj[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) Pe[k] = R[3 * k] * lp[0] + R[3 * k + 1] * lp[1] + R[3 * k + 2] * lp[2] + a.anc[k];
                double d_u[3], d_r[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { d_u[k] = F.user_sat_pos[i][k] - Pe[k]; d_r[k] = F.ref_sat_pos[i][k] - F.station[k]; }
                const double r_u = sqrt(d_dot3_nc(d_u, d_u)), r_r = sqrt(d_dot3_nc(d_r, d_r));
                sRu[fl][i] = r_u; sRr[fl][i] = r_r; sObs[fl][i] = F.user_psr[i] - F.ref_psr[i];
#pragma unroll
                for (int c = 0; c < 3; ++c) sE[fl][i][c] = (d_u[0] * R[c] + d_u[1] * R[3 + c] + d_u[2] * R[6 + c]) / r_u;
            }
        }
        GN_STAMP(1);
        // From here to the per-factor parts everything a factor needs was produced by ITS OWN 32 lanes (one half of a wavefront):
        // wavefront-level LDS synchronisation is enough, the workgroup barrier comes only before the factors are added up.
        GLIO_WAVE_LDS_SYNC();
        GN_STAMP(2);
        if (!ddt_have) { if (tid < dop_cnt0) ddt_p = x[16 * W + rowp.epoch]; ddt_have = true; }
        if (fl < nf) {
            const int ns = s_nw[fl] + 1, m = s_m[fl];
            if (i < ns && i != m) {
                const int ri = i < m ? i : i - 1;
                const double est = (sRu[fl][i] - sRr[fl][i]) - (sRu[fl][m] - sRr[fl][m]);
                const double obs = sObs[fl][i] - sObs[fl][m];
                const double wgt = fabs(est - obs) > f_thr ? 0.05 : 1.0;     // :99-102
                raw[fl][ri] = wgt * (est - obs);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double ei = sE[fl][i][c], em = sE[fl][m][c];
                    Jri[fl][ri * 3 + c] = (-ei * wgt * f_ratio) - (-em * wgt * f_ratio);
                    Jrj[fl][ri * 3 + c] = (-ei * wgt * (1.0 - f_ratio)) - (-em * wgt * (1.0 - f_ratio));
                }
            }
        }
        GLIO_WAVE_LDS_SYNC();
        if (fl < nf && i < s_nw[fl]) {        // residual = W r, J = W J  (:151-167)
            const int nw = s_nw[fl];
            double sr = 0, s6[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < nw; ++b) {
                const double wv = sWt[fl][i * nw + b];
                sr += wv * raw[fl][b];
#pragma unroll
                for (int k = 0; k < 3; ++k) { s6[k] += wv * Jri[fl][b * 3 + k]; s6[3 + k] += wv * Jrj[fl][b * 3 + k]; }
            }
            wE[fl][i * 8 + 6] = sr;
#pragma unroll
            for (int k = 0; k < 6; ++k) wE[fl][i * 8 + k] = s6[k];
        }
        GLIO_WAVE_LDS_SYNC();
        // the factor's own 32 lanes take its 43 entries (36 of H6, 6 of g6, the cost): each a dot product of two columns of the factor's
        // whitened rows; then 43 lanes add the factors' parts in factor order (the association of the sequential formulation: rows
        // inside a factor first, factors after)
        // (H6 is symmetric and its two halves are the same products added in the same order: 21 + 6 + 1 = 28 UNIQUE entries, one per lane -- 43 entries
        //  over 32 lanes were two rounds of the row loop)
        if (fl < nf && i < 28) {
            int ua, ub;
            if (i < 21) { ua = 0; int rem = i; while (rem >= 6 - ua) { rem -= 6 - ua; ++ua; } ub = ua + rem; }      // upper triangle, row by row
            else if (i < 27) { ua = i - 21; ub = 6; }
            else { ua = 6; ub = 6; }
            const int nw = s_nw[fl];
            double sacc = 0;
            for (int r2 = 0; r2 < nw; ++r2) sacc += wE[fl][r2 * 8 + ua] * wE[fl][r2 * 8 + ub];
            lds_.dpart[fl][i] = sacc;
        }
        __syncthreads();
        if (tid < 43) {
            int e;
            if (tid < 36) { const int r = tid / 6, c2 = tid - 6 * r, lo = min(r, c2), hi = max(r, c2); e = lo * 6 - lo * (lo - 1) / 2 + (hi - lo); }
            else if (tid < 42) e = 21 + (tid - 36);
            else e = 27;
            for (int q = 0; q < nf; ++q) {
                const double sacc = lds_.dpart[q][e];
                if (tid < 36) h6 += sacc; else if (tid < 42) g6 += sacc; else cost_dd += 0.5 * sacc;
            }
        }
        __syncthreads();
    }

    GN_STAMP(3);
    // ---- Doppler rows (dopp_factor.hpp:24-75 + HuberLoss(1.0), Estimator.cpp:2335): all epochs of the pair side by
    //      side, one lane per row; sums are taken epoch by epoch (run = the rows of one epoch, contiguous)
    const double OMG = 7.2921151467e-5, CLIGHT = 2.99792458e8;
    double carry = 0.0;          // tid 192..247: partial c / h / g of an epoch that straddles a chunk boundary
    int carry_run = -1;
    for (int c0 = gr.dop_begin; c0 < gr.dop_end; c0 += DOP_CHUNK) {
        const int cnt = min(DOP_CHUNK, gr.dop_end - c0);
        if (tid < cnt) {
            const bool first = c0 == gr.dop_begin;
            const DopRow rw = first ? rowp : load_row(a.dop[c0 + tid]);
            const struct { double ratio, var, sv_ddt, doppler, lamda; double sat_pos[3], sat_vel[3], lever_arm[3]; int epoch; } F =
                {rw.ratio, rw.var, rw.sv_ddt, rw.doppler, rw.lamda, {rw.p0, rw.p1, rw.p2}, {rw.v0, rw.v1, rw.v2}, {rw.l0, rw.l1, rw.l2}, rw.epoch};
            const double Rf[9] = {rw.R0, rw.R1, rw.R2, rw.R3, rw.R4, rw.R5, rw.R6, rw.R7, rw.R8};
            double lp[3], lv[3], Pe[3], Ve[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lp[k] = F.ratio * Pi[k] + (1.0 - F.ratio) * Pj[k] + F.lever_arm[k];
                lv[k] = F.ratio * Vi[k] + (1.0 - F.ratio) * Vj[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pe[k] = Rf[3 * k] * lp[0] + Rf[3 * k + 1] * lp[1] + Rf[3 * k + 2] * lp[2] + a.anc[k];
                Ve[k] = Rf[3 * k] * lv[0] + Rf[3 * k + 1] * lv[1] + Rf[3 * k + 2] * lv[2];
            }
            const double d[3] = {F.sat_pos[0] - Pe[0], F.sat_pos[1] - Pe[1], F.sat_pos[2] - Pe[2]};
            const double rho = sqrt(d_dot3_nc(d, d));
            const double eh[3] = {d[0] / rho, d[1] / rho, d[2] / rho};
            const double sag = OMG / CLIGHT * (F.sat_vel[0] * Pe[1] + F.sat_pos[0] * Ve[1] - F.sat_vel[1] * Pe[0] - F.sat_pos[1] * Ve[0]);
            const double av[3] = {F.sat_vel[0] - Ve[0], F.sat_vel[1] - Ve[1], F.sat_vel[2] - Ve[2]};
            const double ae = d_dot3_nc(av, eh);
            const double ddt = (first && ddt_have) ? ddt_p : x[16 * W + F.epoch];
            const double res = (ae + sag + ddt - F.sv_ddt + F.doppler * F.lamda) / F.var;
            double gP[3], gV[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { gP[k] = -(av[k] - ae * eh[k]) / rho; gV[k] = -eh[k]; }
            gP[0] += OMG / CLIGHT * (-F.sat_vel[1]); gP[1] += OMG / CLIGHT * F.sat_vel[0];
            gV[0] += OMG / CLIGHT * (-F.sat_pos[1]); gV[1] += OMG / CLIGHT * F.sat_pos[0];
            const double iv = 1.0 / F.var;
            // Huber corrector for a scalar residual: sqrt(rho') on J and r, cost rho/2
            const double ar = fabs(res), ah = a.dop_huber;
            const bool inl = ar <= ah;
            const double sw = inl ? 1.0 : sqrt(ah / ar);
            double* J = dE + tid * 16;
            J[14] = inl ? res * res : 2.0 * ah * ar - ah * ah;
            J[13] = sw * res;
            J[15] = 1.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double gPl = gP[0] * Rf[c] + gP[1] * Rf[3 + c] + gP[2] * Rf[6 + c];
                const double gVl = gV[0] * Rf[c] + gV[1] * Rf[3 + c] + gV[2] * Rf[6 + c];
                J[0 + c] = sw * F.ratio * gPl * iv;
                J[3 + c] = sw * F.ratio * gVl * iv;
                J[6 + c] = sw * (1.0 - F.ratio) * gPl * iv;
                J[9 + c] = sw * (1.0 - F.ratio) * gVl * iv;
            }
            J[12] = sw * iv;
        }
        GN_STAMP(4);
        __syncthreads();
        GN_STAMP(5);
        // Reduction.  tid < 182: dot products of two columns of the row records over ALL rows of the chunk --
        //   (u, v) -> H12;  (u, residual) -> g12;  (rho, 1) -> 2 cost  -- these sums run over every epoch anyway.
        // tid 192..247 (wavefront 3, concurrently): the same per EPOCH (run = the rows of one epoch, contiguous), 14 lanes per epoch --
        //   (u, ddt column) -> coupling c[u], u < 12;  (ddt, ddt) -> h;  (ddt, residual) -> g.
        if (tid < 182) {
            // 91 UNIQUE entries (78 of the symmetric H12 -- (u, v) and (v, u) are the same products in the same order --, 12 of g12, the cost), two lanes
            // each: the sums always were two partial sums over the even and the odd rows added at the end; the even lane takes the even rows, its
            // neighbour the odd ones (157 entries by one lane each read every row twice as long)
            const int e = tid >> 1, half = tid & 1;
            int ua, ub;
            if (e < 78) { ua = 0; int rem = e; while (rem >= 12 - ua) { rem -= 12 - ua; ++ua; } ub = ua + rem; }
            else if (e < 90) { ua = e - 78; ub = 13; }
            else { ua = 14; ub = 15; }
            double sh = 0;
            int r2 = half;
            for (; r2 + 6 < cnt; r2 += 8) {         // four rows' reads in flight
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xa[u] = dE[(r2 + 2 * u) * 16 + ua]; xb[u] = dE[(r2 + 2 * u) * 16 + ub]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) sh += xa[u] * xb[u];
            }
            for (; r2 < cnt; r2 += 2) sh += dE[r2 * 16 + ua] * dE[r2 * 16 + ub];
            const double so = __shfl_xor(sh, 1, 64);
            if (half == 0) { const double sacc = sh + so; dq += e < 90 ? sacc : 0.5 * sacc; }
        } else if (tid >= 192 && tid < 192 + 56) {
            // four epochs side by side, 14 lanes each; an epoch keeps its lanes from chunk to chunk (rl is its index in the group), so a
            // partial sum carried over a chunk boundary stays with the lane that continues it
            // (five epochs on lanes 182..251: slower -- wavefront 2 then walks both branches)
            const int slot = (tid - 192) / 14, u = (tid - 192) - 14 * slot;     // u 0..11: coupling, 12: h, 13: g
            const int ua = u < 12 ? u : 12, ub = u < 13 ? 12 : 13;
            for (int rl = slot; rl < n_runs; rl += 4) {
                const int rn = gr.run_begin + rl;
                DopRun run;
                if (rl < GN_MAX_RUNS) run = s_runs[rl]; else run = a.runs[rn];
                const int rb = max(run.begin, c0) - c0, re = min(run.end, c0 + cnt) - c0;      // rows of this epoch in the chunk
                if (rb >= re) continue;
                double s0 = 0, s1 = 0;
                int r2 = rb;
                for (; r2 + 8 <= re; r2 += 8) {
                    double xa[8], xb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { xa[q] = dE[(r2 + q) * 16 + ua]; xb[q] = dE[(r2 + q) * 16 + ub]; }
#pragma unroll
                    for (int q = 0; q < 8; q += 2) { s0 += xa[q] * xb[q]; s1 += xa[q + 1] * xb[q + 1]; }
                }
                for (; r2 + 2 <= re; r2 += 2) { s0 += dE[r2 * 16 + ua] * dE[r2 * 16 + ub]; s1 += dE[(r2 + 1) * 16 + ua] * dE[(r2 + 1) * 16 + ub]; }
                if (r2 < re) s0 += dE[r2 * 16 + ua] * dE[r2 * 16 + ub];
                double sacc = s0 + s1;
                if (carry_run == rn) sacc = carry + sacc;
                if (run.end <= c0 + cnt) {          // epoch complete: publish its clock-drift block
                    DdtBlock* D = ddt_out + run.epoch;
                    if (u < 12) D->c[u] = sacc; else if (u == 12) D->h = sacc; else D->g = sacc;
                    if (u == 0) { D->group = gidx; D->used = 1; }
                } else { carry = sacc; carry_run = rn; }
            }
        }
        __syncthreads();
    }

    GN_STAMP(6);
    // ---- scatter the thread-private accumulators into the dense pair block
    const bool pair_H = a.pair_H != 0;           // (uniform) the dense 30 x 30 block is only written for consumers that read it
    const int map12[12] = {0, 1, 2, 6, 7, 8, 15, 16, 17, 21, 22, 23};
    const bool chain = a.chain_src && !a.marg && sj == si + 1;
    // the DD parts (6 x 6, 6, cost) and the Doppler cost meet the Doppler parts through LDS: one barrier
    if (tid < 36) lds_.sH6[tid] = h6;
    else if (tid < 42) lds_.sG6[tid - 36] = g6;
    else if (tid == 42) s_cost[0] = cost_dd;
    if (tid < 182 && !(tid & 1)) { if ((tid >> 1) < 90) lds_.sU[tid >> 1] = dq; else s_cost[1] = dq; }
    if (pair_H) for (int k = tid; k < GLIO_PAIR_DIM * GLIO_PAIR_DIM; k += SF_THREADS) out->H[k] = 0.0;
    __syncthreads();
    if (tid < 144) {
        const int ia = tid / 12, ib = tid - 12 * ia;
        const int R = map12[ia], C = map12[ib];
        // position inside the DD 6-vector (t of i, t of j) or -1
        const int a6 = (ia % 6) < 3 ? (ia / 6) * 3 + ia % 6 : -1, b6 = (ib % 6) < 3 ? (ib / 6) * 3 + ib % 6 : -1;
        const int lo = min(ia, ib), hi = max(ia, ib);
        double v = lds_.sU[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];
        if (a6 >= 0 && b6 >= 0) v += lds_.sH6[a6 * 6 + b6];          // Doppler 12 x 12 first, DD 6 x 6 added on top
        if (pair_H) out->H[R * GLIO_PAIR_DIM + C] = v;
        if (chain) {
            // chain-layout slices of the two keyframes (only pairs (i, i + 1) take part in the keyframe chain).  Only the 12 x 12 positions
            // of (t v of i, t v of j) can be non-zero; everything else in a GNSS slice was zeroed when the structure was set
            const ChainKf* kd = reinterpret_cast<const ChainKf*>(a.chain_tabs);
            if (R < 15) { if (C <= R) chain_slice(a, which, si, kd[si].k0 == gidx ? 2 : 3)[R * (R + 1) / 2 + C] = v; }                 // aa
            else if (C < 15) chain_slice(a, which, si, kd[si].k0 == gidx ? 2 : 3)[120 + (R - 15) * 15 + C] = v;                       // ba
            else if (C <= R) chain_slice(a, which, sj, kd[sj].k0 == gidx ? 2 : 3)[(R - 15) * (R - 14) / 2 + (C - 15)] = v;             // bb
        }
    } else if (tid < 156) {
        const int ia = tid - 144;
        const int a6 = (ia % 6) < 3 ? (ia / 6) * 3 + ia % 6 : -1;
        const double g12 = lds_.sU[78 + ia];
        out->g[map12[ia]] = a6 >= 0 ? g12 + lds_.sG6[a6] : g12;
    } else if (tid >= 160 && tid < 160 + GLIO_PAIR_DIM) {
        const int k = tid - 160;                     // the 18 entries of g outside (t v of i, t v of j) are zero
        bool in12 = false;
#pragma unroll
        for (int q = 0; q < 12; ++q) in12 = in12 || map12[q] == k;
        if (!in12) out->g[k] = 0.0;
    } else if (tid == 200) { out->cost = s_cost[0] + s_cost[1]; out->slot_a = si; out->slot_b = sj; }
    GN_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// Marginalization prior (MarginalizationFactor.cpp:233-287)
//   r = r0 + J0 dx ; local Jacobian = J0 * blockdiag(M_b), M_b = I or 2 s Qleft(q0^-1)[1:4,:] P(q)
//   H = M^T (J0^T J0) M, g = M^T J0^T r, cost = |r|^2/2
// ------------------------------------------------------------------------------------------------
#define PRIOR_H_BLOCKS 24     // workgroups that share the H = M^T A0 M entries; one more does r, g, cost

// dx [np] and the 3x3 blocks M_b of the quaternion blocks, into LDS (every prior workgroup recomputes them)
__device__ __forceinline__ void prior_dx_M(const SmallArgs& a, const double* __restrict__ x, double* dx, double* Mb) {
    const int tid = threadIdx.x, nb = a.npb, W = a.W;
    for (int b = tid; b < nb; b += SF_THREADS) {
        const int s = a.pslot[b], kind = a.pkind[b], idx = a.pidx[b];
        const double* x0 = a.px0 + 9 * b;
        if (kind == GLIO_BLK_TRANS) {
            for (int k = 0; k < 3; ++k) dx[idx + k] = x[3 * s + k] - x0[k];
        } else if (kind == GLIO_BLK_SPEEDBIAS) {
            for (int k = 0; k < 9; ++k) dx[idx + k] = x[7 * W + 9 * s + k] - x0[k];
        } else {
            double q[4], q0inv[4], dq[4];
            for (int k = 0; k < 4; ++k) q[k] = x[3 * W + 4 * s + k];
            d_qinv(x0, q0inv);
            d_qmul(q0inv, q, dq);
            const double sg = dq[0] >= 0 ? 2.0 : -2.0;           // :246-252, :276-281
            d_qnormalize(dq);
            for (int k = 0; k < 3; ++k) dx[idx + k] = sg * dq[1 + k];
            double L[16], P[12];
            qleft16(q0inv, L);
            d_plus_jac(q, P);
            for (int p = 0; p < 3; ++p)
                for (int c = 0; c < 3; ++c) {
                    double acc = 0;
                    for (int k = 0; k < 4; ++k) acc += L[(1 + p) * 4 + k] * P[k * 3 + c];
                    Mb[9 * b + p * 3 + c] = a.marg ? sg * L[(1 + p) * 4 + 1 + c] : sg * acc;
                }
        }
    }
    __syncthreads();
}

#define PRIOR_MAX_NP (6 * GLIO_MAX_WINDOW + 9)
#define PRIOR_MAX_NB (2 * GLIO_MAX_WINDOW + 1)

// workgroup 0 of the prior: r = r0 + J0 dx (one wavefront per row, coalesced), v = J0^T r, g = M^T v, cost
struct PriorLds { double dx[PRIOR_MAX_NP], r[PRIOR_MAX_NP], v[PRIOR_MAX_NP], Mb[9 * PRIOR_MAX_NB]; };
union SmallLds { ImuLds imu; GnssLds gn; PriorLds pr; };
__device__ void prior_rg_block(const SmallArgs& a, const double* __restrict__ x, double* gout, double* cost, unsigned char* pool) {
    PriorLds& lds_ = *reinterpret_cast<PriorLds*>(pool);
    double (&dx)[PRIOR_MAX_NP] = lds_.dx; double (&r)[PRIOR_MAX_NP] = lds_.r; double (&v)[PRIOR_MAX_NP] = lds_.v; double (&Mb)[9 * PRIOR_MAX_NB] = lds_.Mb;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, np = a.np;
    prior_dx_M(a, x, dx, Mb);
    // r = r0 + J0 dx: one lane per row, four independent accumulators -- the loads of a row do not depend on each other, so
    // the whole product costs a few latency rounds instead of one wave-reduction round trip per row
    for (int i = tid; i < np; i += SF_THREADS) {
        const double* row = a.pJ0 + (size_t)i * np;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = 0;
        for (; k + 4 <= np; k += 4) { s0 += row[k] * dx[k]; s1 += row[k + 1] * dx[k + 1]; s2 += row[k + 2] * dx[k + 2]; s3 += row[k + 3] * dx[k + 3]; }
        for (; k < np; ++k) s0 += row[k] * dx[k];
        r[i] = a.pr0[i] + ((s0 + s1) + (s2 + s3));
    }
    (void)lane; (void)wv;
    __syncthreads();
    for (int j = tid; j < np; j += SF_THREADS) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int i = 0;
        for (; i + 4 <= np; i += 4) {
            s0 += a.pJ0[(size_t)i * np + j] * r[i]; s1 += a.pJ0[(size_t)(i + 1) * np + j] * r[i + 1];
            s2 += a.pJ0[(size_t)(i + 2) * np + j] * r[i + 2]; s3 += a.pJ0[(size_t)(i + 3) * np + j] * r[i + 3];
        }
        for (; i < np; ++i) s0 += a.pJ0[(size_t)i * np + j] * r[i];
        v[j] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int j = tid; j < np; j += SF_THREADS) {
        const int b = a.pcolblk[j];
        double s;
        if (a.pkind[b] == GLIO_BLK_QUAT) {
            const int idx = a.pidx[b], c = j - idx;
            s = Mb[9 * b + 0 * 3 + c] * v[idx] + Mb[9 * b + 1 * 3 + c] * v[idx + 1] + Mb[9 * b + 2 * 3 + c] * v[idx + 2];
        } else s = v[j];
        gout[j] = s;
    }
    if (tid < 64) {
        double s = 0;
        for (int i = tid; i < np; i += 64) s += r[i] * r[i];
        s = wave_sum(s);
        if (tid == 0) *cost = 0.5 * s;
    }
}

// workgroups 1..PRIOR_H_BLOCKS: rows i = part, part + PRIOR_H_BLOCKS, ... of H = M^T A0 M
__device__ void prior_H_block(const SmallArgs& a, const double* __restrict__ x, double* H, int part, unsigned char* pool, const int which) {
    PriorLds& lds_ = *reinterpret_cast<PriorLds*>(pool);
    double (&dx)[PRIOR_MAX_NP] = lds_.dx; double (&Mb)[9 * PRIOR_MAX_NB] = lds_.Mb;
    const int tid = threadIdx.x, np = a.np;
    prior_dx_M(a, x, dx, Mb);
    for (int i = part; i < np; i += PRIOR_H_BLOCKS) {
        const int bi = a.pcolblk[i];
        const bool qi = a.pkind[bi] == GLIO_BLK_QUAT;
        const int ii = a.pidx[bi], ci = i - ii;
        const int slot_i = a.pslot[bi];
        for (int j = tid; j < np; j += SF_THREADS) {
            const int bj = a.pcolblk[j];
            const bool qj = a.pkind[bj] == GLIO_BLK_QUAT;
            const int jj = a.pidx[bj], cj = j - jj;
            double s;
            if (!qi && !qj) s = a.pA0[(size_t)i * np + j];
            else if (qi && !qj) {
                s = 0;
                for (int k = 0; k < 3; ++k) s += Mb[9 * bi + k * 3 + ci] * a.pA0[(size_t)(ii + k) * np + j];
            } else if (!qi && qj) {
                s = 0;
                for (int l = 0; l < 3; ++l) s += a.pA0[(size_t)i * np + jj + l] * Mb[9 * bj + l * 3 + cj];
            } else {
                s = 0;
                for (int k = 0; k < 3; ++k) {
                    double t = 0;
                    for (int l = 0; l < 3; ++l) t += a.pA0[(size_t)(ii + k) * np + jj + l] * Mb[9 * bj + l * 3 + cj];
                    s += Mb[9 * bi + k * 3 + ci] * t;
                }
            }
            H[(size_t)i * np + j] = s;
            if (a.chain_src && !a.marg) {        // chain layout: same-keyframe entries (lower triangle) and entries towards the keyframe below
                const int sr = slot_i, sc = a.pslot[bj];
                const int lr = (a.pkind[bi] == GLIO_BLK_TRANS ? 0 : (a.pkind[bi] == GLIO_BLK_QUAT ? 3 : 6)) + ci;
                const int lc = (a.pkind[bj] == GLIO_BLK_TRANS ? 0 : (a.pkind[bj] == GLIO_BLK_QUAT ? 3 : 6)) + cj;
                if (sr == sc) { if (lc <= lr) chain_slice(a, which, sr, 4)[lr * (lr + 1) / 2 + lc] = s; }
                else if (sr == sc + 1) chain_slice(a, which, sc, 4)[120 + lr * 15 + lc] = s;
            }
        }
    }
}

__device__ void small_factors_body(const SmallArgs& a, int role);
__global__ __launch_bounds__(SF_THREADS) void k_small_factors(const SmallArgs a) {
    __builtin_amdgcn_s_setprio(3);      // (O(W) latency-bound workgroups; in the keyframe call they run beside the batch association's wide launches)
#ifdef GLIO_DEV_STAMPS
    const long long t0 = wall_clock64();
    small_factors_body(a, (int)blockIdx.x);
    __syncthreads();
    if (threadIdx.x == 0 && a.dbg && blockIdx.x < 250) a.dbg[blockIdx.x] = wall_clock64() - t0;     // scripts/small_time.py
#else
    small_factors_body(a, (int)blockIdx.x);
#endif
}
__device__ void small_factors_body(const SmallArgs& a, const int role) {
    __shared__ __attribute__((aligned(16))) unsigned char pool[sizeof(SmallLds)];
    int which = a.fixed_which;
    if (a.use_status) {
        if (a.st->done || !a.st->cand_pending) return;
        which = 1 - a.st->cur;
    }
    const double* x = which ? a.x1 : a.x0;
    int b = role;                // (the K3 partials are summed by their consumers, k_assemble / k_lidar_reduce: this kernel does not depend on K3)
    if (b < a.n_imu) {
        const int si = a.imu[b].slot_i, sj = si + 1, W = a.W;
        imu_block(a.gravity, x + 3 * si, x + 3 * W + 4 * si, x + 7 * W + 9 * si, x + 3 * sj, x + 3 * W + 4 * sj, x + 7 * W + 9 * sj,
                  a.imu[b], a.imu_blocks + (size_t)which * a.W + b, nullptr, a.marg, pool,
                  a.marg ? nullptr : chain_slice(a, which, si, 0), a.marg ? nullptr : chain_slice(a, which, sj, 1), (b == 0 && a.dbg) ? a.dbg + 190 : nullptr, a.pair_H != 0);
        return;
    }
    b -= a.n_imu;
    if (b < a.n_groups) {
        gnss_block(a, x, a.groups[b], b, a.gnss_blocks + (size_t)which * a.gnss_stride + b, a.ddt_blocks + (size_t)which * a.ddt_stride, pool, which);
        return;
    }
    b -= a.n_groups;
    if (a.has_prior) {
        if (b == 0) prior_rg_block(a, x, a.pg + (size_t)which * a.np, a.pcost + which, pool);
        else prior_H_block(a, x, a.pH + (size_t)which * a.np * a.np, b - 1, pool, which);
    }
}

// ------------------------------------------------------------------------------------------------
// k_linearize_all: K3 and the small factors in ONE launch.  They only share their input (the candidate state), so the
// workgroups of both run side by side: the first n_small workgroups take the small-factor roles (long, latency-bound
// chains on few CUs), the rest are K3 workgroups streaming the correspondences on the others.  The launch lasts about as
// long as the slower of the two instead of their sum.  (Two streams with fork/join events cost more than they save on
// this runtime, and hipExtAnyOrderLaunch is ignored on gfx9; one heterogeneous launch needs neither.)
// ------------------------------------------------------------------------------------------------
struct K3Args {
    const float4* pts; const float4* planes; const double* scores; const int* count; int cap;
    LidarConst lc; double* partials; size_t pstride; int n_small, nb, skip_lo, skip_hi, n_k3;
};
template <bool F32>
__global__ __launch_bounds__(SF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linearize_all(const SmallArgs a, const K3Args k) {
    static_assert(SF_THREADS == GLIO_K3_THREADS, "one block size for both roles");
    __shared__ __attribute__((aligned(16))) float k3f_tile[F32 ? (GLIO_K3_THREADS / GLIO_WAVE) * K3F_TILE_FLOATS : 4];
    __shared__ double k3f_red[F32 ? (GLIO_K3_THREADS / GLIO_WAVE) * 72 : 1];
    int b = (int)blockIdx.x;
    const int role = b < k.n_small ? b : -1;
#ifdef GLIO_DEV_STAMPS
    if (role >= 0) {          // per-workgroup duration of the small-factor roles (scripts/small_time.py)
        const long long t0 = wall_clock64();
        small_factors_body(a, role);
        __syncthreads();
        if (threadIdx.x == 0 && a.dbg && role < 190) a.dbg[role] = wall_clock64() - t0;
        if (threadIdx.x == 0 && a.dbg && role == 0) a.dbg[197] = t0;
        return;
    }
    // (end of the K3 workgroups: plain stores into 32 slots by workgroup index -- the last writers are the last to finish; no atomics, which would
    //  serialise at the memory side and lengthen what they measure)
    struct K3End { long long* d; __device__ ~K3End() { __syncthreads(); if (threadIdx.x == 0 && d) d[208 + (blockIdx.x & 31)] = wall_clock64(); } } k3end{a.dbg};
#else
    if (role >= 0) { small_factors_body(a, role); return; }
#endif
    int which = a.fixed_which;
    if (a.use_status) {
        if (a.st->done || !a.st->cand_pending) return;
        which = 1 - a.st->cur;
    }
    // workgroups [skip_lo, skip_hi) stay idle: in dispatch order they would land in the second slot of the CUs that run the
    // small-factor workgroups and slow those (the launch's long pole) down
    // (the small-factor workgroups two to a CU, [0, n/2) and [256, 256 + n/2), and every other slot for K3 -- 22 instead of 19 workgroups per keyframe:
    //  16.2 us against 12.9; with K3 workgroups in those slots 15.7.  A small-factor workgroup wants its CU alone.)
    if (b >= k.skip_lo && b < k.skip_hi) return;
    b -= k.n_small + (b >= k.skip_hi ? k.skip_hi - k.skip_lo : 0);
    if (b >= k.n_k3) return;
    const int kf = b / k.nb, bx = b - kf * k.nb;
    double* part = k.partials + (size_t)which * k.pstride;
    if (F32) k3_body_f32<true>(k.pts, k.planes, k.count, k.cap, which ? a.x1 : a.x0, a.W, k.lc, part, kf, bx, k.nb, k3f_tile, k3f_red);
    else k3_body<2, false, true, true>(k.pts, k.planes, k.scores, k.count, k.cap, which ? a.x1 : a.x0, a.W, k.lc, part, kf, bx, k.nb);
}

// fixed-order sum of the K3 partials of keyframe blockIdx.x into its 28-double block (consumers that want the blocks
// themselves: the marginalization's assembly; k_assemble sums the partials on the fly instead)
__global__ __launch_bounds__(64) void k_lidar_reduce(const double* __restrict__ partials, const int nb, double* __restrict__ blocks) {
    if (threadIdx.x >= GLIO_LIDAR_ACC) return;
    const double* p = partials + (size_t)blockIdx.x * nb * GLIO_LIDAR_ACC + threadIdx.x;
    double s = 0;
    for (int k0 = 0; k0 < nb; k0 += 8) {           // eight independent loads in flight, added in index order
        double v8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v8[q] = p[(size_t)(k0 + q < nb ? k0 + q : k0) * GLIO_LIDAR_ACC];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += k0 + q < nb ? v8[q] : 0.0;
    }
    blocks[(size_t)blockIdx.x * GLIO_LIDAR_ACC + threadIdx.x] = s;
}
void glio_launch_lidar_reduce(glio_ctx* c, int which) {
    hipLaunchKernelGGL(k_lidar_reduce, dim3(c->W), dim3(64), 0, c->stream, c->d_lidar_partials + (size_t)which * glio_partials_stride(c), c->last_k3_nb,
                       c->d_lidar_blocks + (size_t)which * c->W * GLIO_LIDAR_ACC);
}

// ------------------------------------------------------------------------------------------------
// assemble: gather every block into the dense normal equations (one thread per H entry)
// ------------------------------------------------------------------------------------------------
struct AsmArgs {
    int W, n, n_ddt, n_imu, n_groups, has_prior, np;
    int band;                 // 1: pose rows are written in the block-tridiagonal band and the epoch columns only (k_chain_solve<true> reads nothing else)
    const SolverStatus* st; int use_status; int fixed_which;
    const double* lidar_partials; size_t lidar_pstride; int lidar_nb; const PairBlock* imu_blocks; const PairBlock* gnss_blocks; const DdtBlock* ddt_blocks;
    int gnss_stride, ddt_stride;
    const double* pH; const double* pg; const double* pcost; const int* prior_index;
    double* H0; double* H1; double* g0; double* g1; double* c0; double* c1;
};

__device__ __forceinline__ int lidar_sym_index(int i, int j) {   // upper-triangle packed index, i<=j<6
    return i * 6 - (i * (i - 1)) / 2 + (j - i);
}
__device__ __forceinline__ int dop_local12(int slot_is_j, int lc) {  // pose-local column -> 12-vector index or -1
    int k;
    if (lc < 3) k = lc; else if (lc >= 6 && lc < 9) k = 3 + (lc - 6); else return -1;
    return slot_is_j ? 6 + k : k;
}

// Every entry of H sums up to six block entries that live in different arrays.  All addresses come from LDS lookup tables
// (built once per workgroup from one round of coalesced global reads) and every candidate is loaded UNCONDITIONALLY from a
// clamped, always valid address and selected afterwards, so that a thread has all its global loads in flight together:
// one memory round trip per entry instead of one per `if`.  The terms are added in a fixed order (LiDAR, IMU edge
// (lo, lo+1), IMU edge (sr-1, sr), GNSS groups by ascending index, prior) whatever is present.
__global__ __launch_bounds__(1024) void k_assemble(const AsmArgs a) {
    int which = a.fixed_which;
    if (a.use_status) {
        if (a.st->done || !a.st->cand_pending) return;
        which = 1 - a.st->cur;
    }
    double* H = which ? a.H1 : a.H0;
    double* g = which ? a.g1 : a.g0;
    const int n = a.n, W = a.W, np15 = 15 * W;
    const PairBlock* imu = a.imu_blocks + (size_t)which * W;
    const PairBlock* gn = a.gnss_blocks + (size_t)which * a.gnss_stride;
    const DdtBlock* dd = a.ddt_blocks + (size_t)which * a.ddt_stride;
    // LiDAR blocks: the fixed-order sum over the keyframe's K3 partials is taken right here, all nb loads of an entry in
    // flight together (only the 6 x 6 pose entries and the 6 gradient entries of a keyframe have a LiDAR term)
    const double* lp = a.lidar_partials + (size_t)which * a.lidar_pstride;
    const int lnb = a.lidar_nb;
    auto lidar_term = [&](const int slot, const int idx, const bool live) {
        double s = 0;
        if (live) {
            const double* p = lp + (size_t)slot * lnb * GLIO_LIDAR_ACC + idx;
            for (int k0 = 0; k0 < lnb; k0 += 48) {          // 48 loads in flight per round (one round at the default geometry)
                double vb[48];
#pragma unroll
                for (int q = 0; q < 48; ++q) vb[q] = p[(size_t)(k0 + q < lnb ? k0 + q : k0) * GLIO_LIDAR_ACC];
#pragma unroll
                for (int q = 0; q < 48; ++q) s += k0 + q < lnb ? vb[q] : 0.0;
            }
        }
        return s;
    };
    const double* pH = a.pH + (size_t)which * a.np * a.np;
    const double* pg = a.pg + (size_t)which * a.np;
    __shared__ short imap[GLIO_MAX_WINDOW], isb[GLIO_MAX_WINDOW];                  // IMU edge that starts at a slot, its other slot
    __shared__ short gmap[GLIO_MAX_WINDOW * GLIO_MAX_WINDOW];                      // GNSS group of a slot pair
    __shared__ short gsa[GLIO_MAX_WINDOW * GLIO_MAX_WINDOW], gsb[GLIO_MAX_WINDOW * GLIO_MAX_WINDOW];   // slots of every GNSS group
    __shared__ short gadj0[GLIO_MAX_WINDOW], gadj1[GLIO_MAX_WINDOW], gnext[GLIO_MAX_WINDOW];   // first two groups touching a slot; where a third may start
    __shared__ short pidx[15 * GLIO_MAX_WINDOW];                                   // prior row of a pose parameter
    for (int k = threadIdx.x; k < W; k += blockDim.x) { imap[k] = -1; isb[k] = -1; }
    for (int k = threadIdx.x; k < W * W; k += blockDim.x) gmap[k] = -1;
    for (int k = threadIdx.x; k < np15; k += blockDim.x) pidx[k] = a.has_prior ? (short)a.prior_index[k] : (short)-1;
    __syncthreads();
    for (int k = threadIdx.x; k < a.n_imu; k += blockDim.x) { const int sa = imu[k].slot_a; imap[sa] = (short)k; isb[sa] = (short)imu[k].slot_b; }
    for (int k = threadIdx.x; k < a.n_groups; k += blockDim.x) {
        const int sa = gn[k].slot_a, sb = gn[k].slot_b;
        gsa[k] = (short)sa; gsb[k] = (short)sb;
        gmap[sa * W + sb] = (short)k;
        gmap[sb * W + sa] = (short)k;
    }
    __syncthreads();
    for (int sl = threadIdx.x; sl < W; sl += blockDim.x) {
        int k0 = -1, k1 = -1, nx = a.n_groups;
        for (int k = 0; k < a.n_groups; ++k)
            if (gsa[k] == sl || gsb[k] == sl) {
                if (k0 < 0) k0 = k; else if (k1 < 0) k1 = k; else { nx = k; break; }
            }
        gadj0[sl] = (short)k0; gadj1[sl] = (short)k1; gnext[sl] = (short)nx;
    }
    __syncthreads();
    // rows are dealt to workgroups, columns to lanes (coalesced stores, no integer division per entry).  Band mode: a row has ~45 + n_ddt live
    // columns, so a workgroup of 1024 takes eight rows at a time, 128 lanes each.
    const int r_first = a.band ? 8 * blockIdx.x + (threadIdx.x >> 7) : blockIdx.x, r_stride = a.band ? 8 * gridDim.x : gridDim.x;
    const int c_first = a.band ? (threadIdx.x & 127) : threadIdx.x, c_stride = a.band ? 128 : blockDim.x;
    for (int r = r_first; r <= n; r += r_stride) {
        if (r == n) {          // gradient
            for (int c = c_first; c < n; c += c_stride) {
                double s = 0;
                if (c < np15) {
                    const int sc = c / 15, lc = c % 15;
                    const bool lid = lc < 6;
                    const int e0 = imap[sc], e1r = sc > 0 ? imap[sc - 1] : -1;
                    const bool ok1 = e1r >= 0 && isb[sc - (sc > 0)] == sc;
                    const int k0 = gadj0[sc], k1 = gadj1[sc], pi = pidx[c];
                    const double v0 = imu[e0 >= 0 ? e0 : 0].g[lc];
                    const double v1 = imu[ok1 ? e1r : 0].g[15 + lc];
                    const double vg0 = gn[k0 >= 0 ? k0 : 0].g[(k0 >= 0 && gsa[k0] == sc ? 0 : 15) + lc];
                    const double vg1 = gn[k1 >= 0 ? k1 : 0].g[(k1 >= 0 && gsa[k1] == sc ? 0 : 15) + lc];
                    const double vp = pg[pi >= 0 ? pi : 0];
                    const double vl = lidar_term(sc, 21 + (lid ? lc : 0), lid);
                    s += lid ? vl : 0.0;
                    s += e0 >= 0 ? v0 : 0.0;
                    s += ok1 ? v1 : 0.0;
                    s += k0 >= 0 ? vg0 : 0.0;
                    s += k1 >= 0 ? vg1 : 0.0;
                    for (int k = gnext[sc]; k < a.n_groups; ++k) {
                        if (gsa[k] == sc) s += gn[k].g[lc];
                        else if (gsb[k] == sc) s += gn[k].g[15 + lc];
                    }
                    s += pi >= 0 ? vp : 0.0;
                } else s = dd[c - np15].g;
                g[c] = s;
            }
            continue;
        }
        double* Hrow = H + (size_t)r * n;
        if (r < np15) {
            const int sr = r / 15, lr = r % 15;
            const int pi = pidx[r];
            for (int c = c_first; c < n; c += c_stride) {
                double s = 0;
                if (a.band && c < np15 && (c < 15 * (sr - 1) || c >= 15 * (sr + 2))) continue;       // (zero by the chain structure; zeroed once by the host)
                if (c < np15) {
                    const int sc = c / 15, lc = c % 15;
                    const int d = sc - sr;
                    // LiDAR: 6 x 6 pose block of the keyframe
                    const bool lid = (d == 0) & (lr < 6) & (lc < 6);
                    const int lix = lid ? (lr <= lc ? lidar_sym_index(lr, lc) : lidar_sym_index(lc, lr)) : 0;
                    // IMU edge (lo, lo + 1) when both slots belong to it; edge (sr - 1, sr) also covers the block (sr, sr)
                    const int lo = sr < sc ? sr : sc;
                    const bool near = (d >= -1) & (d <= 1);
                    const int e0 = near ? imap[lo] : -1;
                    const int sb0 = near ? isb[lo] : -1;
                    const bool ok0 = (e0 >= 0) & ((sr == lo) | (sr == sb0)) & ((sc == lo) | (sc == sb0));
                    const int off0 = ((sr == lo ? 0 : 15) + lr) * GLIO_PAIR_DIM + (sc == lo ? 0 : 15) + lc;
                    const int e1 = (d == 0 && sr > 0) ? imap[sr - 1] : -1;
                    const bool ok1 = e1 >= 0 && isb[sr - 1] == sr;
                    // GNSS: off-diagonal slot block = the group of the pair; diagonal slot block = every group touching sr
                    int k0, k1, offg0, offg1 = 0;
                    if (d != 0) {
                        k0 = gmap[sr * W + sc]; k1 = -1;
                        const int ga = k0 >= 0 ? gsa[k0] : -1;
                        offg0 = ((sr == ga ? 0 : 15) + lr) * GLIO_PAIR_DIM + (sc == ga ? 0 : 15) + lc;
                    } else {
                        k0 = gadj0[sr]; k1 = gadj1[sr];
                        offg0 = (k0 >= 0 && gsa[k0] == sr) ? lr * GLIO_PAIR_DIM + lc : (15 + lr) * GLIO_PAIR_DIM + 15 + lc;
                        offg1 = (k1 >= 0 && gsa[k1] == sr) ? lr * GLIO_PAIR_DIM + lc : (15 + lr) * GLIO_PAIR_DIM + 15 + lc;
                    }
                    const int pj = pidx[c];
                    const bool okp = (pi >= 0) & (pj >= 0);
                    const double v0 = imu[ok0 ? e0 : 0].H[ok0 ? off0 : 0];
                    const double v1 = imu[ok1 ? e1 : 0].H[(15 + lr) * GLIO_PAIR_DIM + 15 + lc];
                    const double vg0 = gn[k0 >= 0 ? k0 : 0].H[offg0];
                    const double vg1 = gn[k1 >= 0 ? k1 : 0].H[offg1];
                    const double vp = pH[okp ? (size_t)pi * a.np + pj : 0];
                    const double vl = lidar_term(sr, lix, lid);
                    s += lid ? vl : 0.0;
                    s += ok0 ? v0 : 0.0;
                    s += ok1 ? v1 : 0.0;
                    s += k0 >= 0 ? vg0 : 0.0;
                    s += k1 >= 0 ? vg1 : 0.0;
                    if (d == 0)
                        for (int k = gnext[sr]; k < a.n_groups; ++k) {                 // a slot with more than two GNSS groups (general graphs)
                            if (gsa[k] == sr) s += gn[k].H[lr * GLIO_PAIR_DIM + lc];
                            else if (gsb[k] == sr) s += gn[k].H[(15 + lr) * GLIO_PAIR_DIM + 15 + lc];
                        }
                    s += okp ? vp : 0.0;
                } else {
                    const int ep = c - np15;
                    const int used = dd[ep].used, gi0 = dd[ep].group;
                    const int gi = used ? gi0 : 0;
                    const int sa = gsa[gi], sb = gsb[gi];
                    const int k12 = dop_local12(sr == sb && sr != sa, lr);
                    const bool ok = (used != 0) & ((sr == sa) | (sr == sb)) & (k12 >= 0);
                    const double v = dd[ep].c[ok ? k12 : 0];
                    s = ok ? v : 0.0;
                }
                Hrow[c] = s;
            }
        } else {
            const int ep = r - np15;
            const bool used = dd[ep].used != 0;
            const int gi = used ? dd[ep].group : 0;
            const int sa = used ? gsa[gi] : -1, sb = used ? gsb[gi] : -1;
            for (int c = c_first; c < n; c += c_stride) {
                double s = 0;
                if (c >= np15) { if (c == r) s = dd[ep].h; }
                else if (used) {
                    const int sc = c / 15, lc = c % 15;
                    if (sc == sa || sc == sb) { const int k12 = dop_local12(sc == sb && sc != sa, lc); if (k12 >= 0) s = dd[ep].c[k12]; }
                }
                Hrow[c] = s;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {       // total cost: every term fetched by its own lane, fixed-shape wave reduction
        const int l = threadIdx.x;
        double cs = 0;
        for (int k = l; k < W; k += 64) cs += lidar_term(k, 27, true);
        for (int k = l; k < a.n_imu; k += 64) cs += imu[k].cost;
        for (int k = l; k < a.n_groups; k += 64) cs += gn[k].cost;
        if (l == 0 && a.has_prior) cs += a.pcost[which];
        cs = wave_sum(cs);
        if (l == 0) *(which ? a.c1 : a.c0) = cs;
    }
}

// ------------------------------------------------------------------------------------------------
// single-factor evaluators (Ceres Evaluate() convention, GLOBAL Jacobians) for the C-ABI shims
// ------------------------------------------------------------------------------------------------
// params: Pi3 Qi4 SBi9 Pj3 Qj4 SBj9 packed [32]; out: r[15] then J[15][32]
__global__ __launch_bounds__(SF_THREADS) void k_eval_imu(const double gravity, const ImuEdgeDev* e, const double* params, double* out) {
    __shared__ __attribute__((aligned(16))) unsigned char pool[sizeof(ImuLds)];
    imu_block(gravity, params + 0, params + 3, params + 7, params + 16, params + 19, params + 23, *e, nullptr, out, 0, pool);
}

// LidarPlaneNormFactor residual + global 1x3 / 1x4 Jacobians through Eigen's q*v formula
// (LidarKeyframeFactor.h:87-103).  in: t3 q4 ; out: r, Jt[3], Jq[4]
struct LidarEvalArgs { double qlb[4], tlb[3], cp[3], n[3], d, score; };
__global__ void k_eval_lidar(const LidarEvalArgs a, const double* params, double* out) {
    if (threadIdx.x != 0) return;
    const double* t = params;
    const double* q = params + 3;
    double qlbi[4], c[3] = {a.cp[0] - a.tlb[0], a.cp[1] - a.tlb[1], a.cp[2] - a.tlb[2]}, pb[3], pw[3];
    d_qinv(a.qlb, qlbi);
    d_qrot(qlbi, c, pb);
    d_qrot(q, pb, pw);
    for (int k = 0; k < 3; ++k) pw[k] += t[k];
    out[0] = a.score * (d_dot3(a.n, pw) + a.d);
    for (int k = 0; k < 3; ++k) out[1 + k] = a.score * a.n[k];
    const double w = q[0];
    const double* u = q + 1;
    double uv[3];
    d_cross(u, pb, uv);
    out[4] = a.score * 2.0 * d_dot3(a.n, uv);
    const double Sv[9] = {0, -pb[2], pb[1], pb[2], 0, -pb[0], -pb[1], pb[0], 0};
    const double Suv[9] = {0, -uv[2], uv[1], uv[2], 0, -uv[0], -uv[1], uv[0], 0};
    const double Su[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
    for (int k = 0; k < 3; ++k) {
        double col[3];
        for (int r = 0; r < 3; ++r) {
            double susv = Su[r * 3] * Sv[k] + Su[r * 3 + 1] * Sv[3 + k] + Su[r * 3 + 2] * Sv[6 + k];
            col[r] = -2 * w * Sv[r * 3 + k] - 2 * Suv[r * 3 + k] - 2 * susv;
        }
        out[5 + k] = a.score * d_dot3(a.n, col);
    }
}

// The ImuFactor chain of the batch problem (Estimator.cpp:2990-3001; batch_tr_kernels.hip): one workgroup per edge e in
// [e0, e1) between keyframes e and e + 1 of the poses [K][7] / speed-bias [K][9] arrays picked by `sel`; the 30 x 30 local
// J^T J, J^T r and the cost go to rec[e] (the same device code as the sliding window's IMU role).
__global__ __launch_bounds__(SF_THREADS) void k_batch_imu(const BtSel sel, const double gravity, const ImuEdgeDev* __restrict__ edges, const int e0,
                                                          const double* __restrict__ poses0, const double* __restrict__ poses1,
                                                          const double* __restrict__ sb0, const double* __restrict__ sb1, PairBlock* rec0, PairBlock* rec1) {
    __shared__ __attribute__((aligned(16))) unsigned char pool[sizeof(ImuLds)];
    if (bt_skip(sel)) return;
    const int pick = bt_pick(sel), e = e0 + blockIdx.x;
    const double* poses = pick ? poses1 : poses0;
    const double* sb = pick ? sb1 : sb0;
    PairBlock* rec = pick ? rec1 : rec0;
    const double* pi = poses + 7 * (size_t)e;
    const double* pj = poses + 7 * (size_t)(e + 1);
    imu_block(gravity, pi, pi + 3, sb + 9 * (size_t)e, pj, pj + 3, sb + 9 * (size_t)(e + 1), edges[e], rec + e, nullptr, 0, pool);
}
void glio_launch_batch_imu(hipStream_t stream, const BtSel& sel, double gravity, const ImuEdgeDev* edges, int e0, int e1, const double* poses0, const double* poses1,
                           const double* sb0, const double* sb1, PairBlock* rec0, PairBlock* rec1) {
    if (e1 > e0) hipLaunchKernelGGL(k_batch_imu, dim3(e1 - e0), dim3(SF_THREADS), 0, stream, sel, gravity, edges, e0, poses0, poses1, sb0, sb1, rec0, rec1);
}

void glio_launch_eval_imu(glio_ctx* c, const ImuEdgeDev* d_edge, const double* d_params, double* d_out) {
    hipLaunchKernelGGL(k_eval_imu, dim3(1), dim3(SF_THREADS), 0, c->stream, c->opts.gravity, d_edge, d_params, d_out);
}
void glio_launch_eval_lidar(glio_ctx* c, const float cp[4], const float plane[4], double score, const double* d_params, double* d_out) {
    LidarEvalArgs a;
    for (int k = 0; k < 4; ++k) a.qlb[k] = c->opts.q_lb[k];
    for (int k = 0; k < 3; ++k) { a.tlb[k] = c->opts.t_lb[k]; a.cp[k] = (double)cp[k]; a.n[k] = (double)plane[k]; }
    a.d = (double)plane[3]; a.score = score;
    hipLaunchKernelGGL(k_eval_lidar, dim3(1), dim3(64), 0, c->stream, a, d_params, d_out);
}

// A0 = J0^T J0 of the prior (once per glio_set_prior)
__global__ void k_gram(const double* __restrict__ J0, double* __restrict__ A0, int np) {
    __builtin_amdgcn_s_setprio(3);      // (see k_marg_inv)
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)np * np) return;
    const int i = (int)(e / np), j = (int)(e % np);
    double s = 0;
    int k = 0;
    for (; k + 8 <= np; k += 8) {            // eight rows' loads in flight; the sum keeps its row order (bit-identical to the plain loop)
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = J0[(size_t)(k + u) * np + i]; b[u] = J0[(size_t)(k + u) * np + j]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u] * b[u];
    }
    for (; k < np; ++k) s += J0[(size_t)k * np + i] * J0[(size_t)k * np + j];
    A0[e] = s;
}
void glio_launch_gram(glio_ctx* c, int np) {
    const long total = (long)np * np;
    hipLaunchKernelGGL(k_gram, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_prior_J0, c->d_prior_A0, np);
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
GnssDevExtra* glio_extra(glio_ctx* c);   // defined in capi.hip

static int fill_small_args(glio_ctx* c, int use_status_cand, int which, int n_ddt, int marg, SmallArgs& a) {
    a.marg = marg;
    a.dbg = c->arrow.d_dbg + 64;
    GnssDevExtra* ex = glio_extra(c);
    a.W = c->W; a.n_imu = c->n_imu; a.n_groups = c->n_groups; a.has_prior = c->prior_n > 0; a.n_ddt = n_ddt;
    a.x0 = c->d_x[0]; a.x1 = c->d_x[1];
    a.st = c->d_status; a.use_status = use_status_cand; a.fixed_which = which;
    a.imu = c->d_imu; a.imu_blocks = c->d_imu_blocks;
    a.dd = c->d_dd; a.dop = c->d_dop; a.groups = c->d_groups; a.runs = ex->d_runs; a.n_runs = ex->n_runs;
    a.gnss_blocks = c->d_gnss_blocks; a.ddt_blocks = c->d_ddt_blocks; a.gnss_stride = c->W * c->W; a.ddt_stride = c->n_ddt_max > 0 ? c->n_ddt_max : 1;
    a.gravity = c->opts.gravity; a.dop_huber = c->opts.doppler_huber_delta;
    for (int k = 0; k < 9; ++k) a.R_ecef_local[k] = c->R_ecef_local[k];
    for (int k = 0; k < 3; ++k) a.anc[k] = c->frame.anc_ecef[k];
    a.np = c->prior_n; a.npb = c->prior_nb;
    a.pJ0 = c->d_prior_J0; a.pA0 = c->d_prior_A0; a.pr0 = c->d_prior_r0; a.px0 = c->d_prior_x0;
    a.pslot = c->d_prior_slot; a.pkind = c->d_prior_kind; a.pidx = c->d_prior_idx; a.pcolblk = ex->d_prior_colblk;
    a.pH = c->d_prior_H; a.pg = c->d_prior_g; a.pcost = c->d_prior_cost; a.pwork = c->d_prior_work;
    a.chain_src = c->d_chain_src; a.chain_tabs = c->d_chain_tabs; a.pair_H = (marg || !c->d_chain_src) ? 1 : c->want_pair_H;
    return c->n_imu + c->n_groups + (a.has_prior ? 1 + PRIOR_H_BLOCKS : 0);
}
void glio_launch_small_factors(glio_ctx* c, int use_status_cand, int which, int n_ddt, int marg) {
    SmallArgs a;
    const int blocks = fill_small_args(c, use_status_cand, which, n_ddt, marg, a);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_small_factors, dim3(blocks), dim3(SF_THREADS), 0, c->stream, a);
}
// K3 + small factors in one launch.  Two workgroups fit a CU (register file and LDS of the small-factor roles): the K3 share
// of the grid is what is left of 2 x 256 slots after the small-factor workgroups, so that everything is resident at once.
void glio_launch_linearize_all(glio_ctx* c, int use_status_cand, int which, int n_ddt) {
    SmallArgs a;
    const int n_small = fill_small_args(c, use_status_cand, which, n_ddt, 0, a);
    if (n_small == 0) { glio_launch_lidar_linearize(c, use_status_cand, which); return; }
    K3Args k;
    k.pts = c->d_pts; k.planes = c->d_planes; k.scores = c->d_scores; k.count = c->d_count; k.cap = c->cap;
    k.lc = glio_lidar_const(c); k.partials = c->d_lidar_partials; k.pstride = glio_partials_stride(c); k.n_small = n_small;
    const bool skip = c->merged_linearize == 2 && n_small <= 128;
    int nb = (512 - (skip ? 2 : 1) * n_small) / c->W;
    if (nb < 4) nb = 4;
    if (nb > c->k3_bpk) nb = c->k3_bpk;
    {   // large keyframes (C5: 262 144 residuals each) get more workgroups than fit at once -- about 4 k residuals per
        // workgroup; the small-factor workgroups are first in dispatch order and still start with the launch
        int maxn = 0;
        for (int s2 = 0; s2 < c->W; ++s2) if (c->h_count[s2] > maxn) maxn = c->h_count[s2];
        int want = (maxn + 4095) / 4096;
        if (want > GLIO_K3_MAX_BLOCKS_PER_KF) want = GLIO_K3_MAX_BLOCKS_PER_KF;
        if (want > nb) nb = want;
    }
    // (the launch now lasts as long as K3 on the CUs the small-factor workgroups leave it -- 193 of 256: with the small-factor roles left out altogether it
    //  takes the same 12.5 us, and K3 alone on the whole chip 9.8.  More K3 workgroups than free slots do not help: from 20 per keyframe on the late ones land
    //  beside the small-factor workgroups and the launch takes 15.3-15.9 us; 17 per keyframe 13.5.)
    k.nb = nb; k.n_k3 = c->W * nb;
    k.skip_lo = skip ? 256 : 0; k.skip_hi = skip ? 256 + n_small : 0;
    c->last_k3_nb = nb;
    if (c->opts.lidar_precision == GLIO_LIDAR_F32_MFMA) {
        glio_lidar_pack_f32(c);
        k.pts = c->d_pts_s;
        hipLaunchKernelGGL(k_linearize_all<true>, dim3(n_small + k.n_k3 + (k.skip_hi - k.skip_lo)), dim3(SF_THREADS), 0, c->stream, a, k);
        return;
    }
    hipLaunchKernelGGL(k_linearize_all<false>, dim3(n_small + k.n_k3 + (k.skip_hi - k.skip_lo)), dim3(SF_THREADS), 0, c->stream, a, k);
}

void glio_launch_assemble(glio_ctx* c, int use_status_cand, int which, int n_ddt, int band) {
    AsmArgs a;
    a.band = band;
    if (band && c->h_band_clean != 15 * c->W + n_ddt) {
        // first band-only assembly since the structure changed / a full assembly ran: whatever the buffers hold outside the band goes
        // (the dense fallback of the chain kernels reads the whole matrix).  No H of the solve being started exists yet.
        const size_t nn = (size_t)(15 * c->W + n_ddt) * (15 * c->W + n_ddt) * sizeof(double);
        hipMemsetAsync(c->d_H[0], 0, nn, c->stream); hipMemsetAsync(c->d_H[1], 0, nn, c->stream);
        c->h_band_clean = 15 * c->W + n_ddt;               // (the row stride is part of it)
    }
    if (!band) c->h_band_clean = 0;
    a.W = c->W; a.n = 15 * c->W + n_ddt; a.n_ddt = n_ddt; a.n_imu = c->n_imu; a.n_groups = c->n_groups;
    a.has_prior = c->prior_n > 0; a.np = c->prior_n;
    a.st = c->d_status; a.use_status = use_status_cand; a.fixed_which = which;
    a.lidar_partials = c->d_lidar_partials; a.lidar_pstride = glio_partials_stride(c); a.lidar_nb = c->last_k3_nb; a.imu_blocks = c->d_imu_blocks; a.gnss_blocks = c->d_gnss_blocks; a.ddt_blocks = c->d_ddt_blocks;
    a.gnss_stride = c->W * c->W; a.ddt_stride = c->n_ddt_max > 0 ? c->n_ddt_max : 1;
    a.pH = c->d_prior_H; a.pg = c->d_prior_g; a.pcost = c->d_prior_cost; a.prior_index = c->d_prior_index;
    a.H0 = c->d_H[0]; a.H1 = c->d_H[1]; a.g0 = c->d_g[0]; a.g1 = c->d_g[1]; a.c0 = c->d_cost[0]; a.c1 = c->d_cost[1];
    const int blocks = band ? (a.n + 1 + 7) / 8 : a.n + 1;       // one workgroup per row of H (+ one for g); band: eight rows each, side by side
    const int threads = (band || a.n + 1 > 1024) ? 1024 : ((a.n + 1 + 63) / 64) * 64;    // one column per thread: a single round of loads per row
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(threads), 0, c->stream, a);
}
