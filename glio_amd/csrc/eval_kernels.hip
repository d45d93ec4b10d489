// eval_kernels.hip -- single-factor evaluators for the GNSS, prior and batch factors with the exact
// ceres::CostFunction::Evaluate(parameters, residuals, jacobians) pointer convention: row-major
// num_residuals x block_size Jacobians in GLOBAL size, any jacobians[i] may be NULL.  They exist so that a maintainer can
// validate the device arithmetic factor by factor against the stock factors (ceres::GradientChecker), not for
// throughput: one small launch per call, scratch allocated per call.
//   glio_eval_dd_psr        dd_psr_factor_20::Evaluate          (GLIO/include/factors/dd_psr_factor.hpp:25-171)
//   glio_eval_doppler       tcdopplerFactor, analytic Jacobians  (dopp_factor.hpp:24-75)
//   glio_eval_marginalization  MarginalizationFactor::Evaluate  (GLIO/src/MarginalizationFactor.cpp:233-287)
//   glio_eval_binary_plane  BinaryLidarPlaneNormFactor           (LidarKeyframeFactor.h:124-164)
#include <cmath>
#include <cstring>
#include <vector>

#include "glio_device.h"

#define EV_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { glio_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)

// out: res[19] | J_Pi[19][3] | J_Pj[19][3]
__global__ __launch_bounds__(64) void k_eval_dd(const glio_dd_psr* __restrict__ Fp, const double* __restrict__ prm /* Pi3 Pj3 R9 anc3 */, double* out) {
#pragma clang fp contract(off)
    __shared__ double raw[19], Ji[57], Jj[57];
    const glio_dd_psr& F = *Fp;
    const int i = threadIdx.x, ns = F.n_sat, m = F.master, nw = ns - 1;
    const double* Pi = prm; const double* Pj = prm + 3; const double* R = prm + 6; const double* anc = prm + 15;
    if (i < 19) { raw[i] = 0; for (int k = 0; k < 3; ++k) { Ji[i * 3 + k] = 0; Jj[i * 3 + k] = 0; } }
    __syncthreads();
    if (i < ns && i != m) {
        double lp[3], Pe[3];
        for (int k = 0; k < 3; ++k) lp[k] = F.ratio * Pi[k] + (1.0 - F.ratio) * Pj[k];
        for (int k = 0; k < 3; ++k) Pe[k] = R[3 * k] * lp[0] + R[3 * k + 1] * lp[1] + R[3 * k + 2] * lp[2] + anc[k];
        const int ri = i < m ? i : i - 1;
        double d_ui[3], d_um[3], d_ri[3], d_rm[3];
        for (int k = 0; k < 3; ++k) {
            d_ui[k] = F.user_sat_pos[i][k] - Pe[k]; d_um[k] = F.user_sat_pos[m][k] - Pe[k];
            d_ri[k] = F.ref_sat_pos[i][k] - F.station[k]; d_rm[k] = F.ref_sat_pos[m][k] - F.station[k];
        }
        const double r_ui = sqrt(d_dot3_nc(d_ui, d_ui)), r_um = sqrt(d_dot3_nc(d_um, d_um)), r_ri = sqrt(d_dot3_nc(d_ri, d_ri)), r_rm = sqrt(d_dot3_nc(d_rm, d_rm));
        const double est = (r_ui - r_ri) - (r_um - r_rm);
        const double obs = (F.user_psr[i] - F.ref_psr[i]) - (F.user_psr[m] - F.ref_psr[m]);
        const double wgt = fabs(est - obs) > F.threshold ? 0.05 : 1.0;
        raw[ri] = wgt * (est - obs);
        for (int c = 0; c < 3; ++c) {
            const double ei = (d_ui[0] * R[c] + d_ui[1] * R[3 + c] + d_ui[2] * R[6 + c]) / r_ui;
            const double em = (d_um[0] * R[c] + d_um[1] * R[3 + c] + d_um[2] * R[6 + c]) / r_um;
            Ji[ri * 3 + c] = (-ei * wgt * F.ratio) - (-em * wgt * F.ratio);
            Jj[ri * 3 + c] = (-ei * wgt * (1.0 - F.ratio)) - (-em * wgt * (1.0 - F.ratio));
        }
    }
    __syncthreads();
    if (i < 19) {                     // W embedded top-left, rows >= nw are zero (dd_psr_factor.hpp:126-167)
        double sr = 0, si[3] = {0, 0, 0}, sj[3] = {0, 0, 0};
        if (i < nw)
            for (int b = 0; b < nw; ++b) {
                const double wv = F.weight[i * nw + b];
                sr += wv * raw[b];
                for (int k = 0; k < 3; ++k) { si[k] += wv * Ji[b * 3 + k]; sj[k] += wv * Jj[b * 3 + k]; }
            }
        out[i] = sr;
        for (int k = 0; k < 3; ++k) { out[19 + i * 3 + k] = si[k]; out[19 + 57 + i * 3 + k] = sj[k]; }
    }
}

// out: res | J_Pi[3] | J_Vi[3] | J_Pj[3] | J_Vj[3] | d/d ddt
__global__ void k_eval_doppler(const glio_doppler* __restrict__ Fp, const double* __restrict__ prm /* Pi3 Vi3 Pj3 Vj3 ddt anc3 */, double* out) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0) return;
    const glio_doppler& F = *Fp;
    const double OMG = 7.2921151467e-5, CLIGHT = 2.99792458e8;
    const double* Pi = prm; const double* Vi = prm + 3; const double* Pj = prm + 6; const double* Vj = prm + 9;
    const double ddt = prm[12];
    const double* anc = prm + 13;
    const double* Rf = F.R_ecef_local;
    double lp[3], lv[3], Pe[3], Ve[3];
    for (int k = 0; k < 3; ++k) { lp[k] = F.ratio * Pi[k] + (1.0 - F.ratio) * Pj[k] + F.lever_arm[k]; lv[k] = F.ratio * Vi[k] + (1.0 - F.ratio) * Vj[k]; }
    for (int k = 0; k < 3; ++k) {
        Pe[k] = Rf[3 * k] * lp[0] + Rf[3 * k + 1] * lp[1] + Rf[3 * k + 2] * lp[2] + anc[k];
        Ve[k] = Rf[3 * k] * lv[0] + Rf[3 * k + 1] * lv[1] + Rf[3 * k + 2] * lv[2];
    }
    const double d[3] = {F.sat_pos[0] - Pe[0], F.sat_pos[1] - Pe[1], F.sat_pos[2] - Pe[2]};
    const double rho = sqrt(d_dot3_nc(d, d));
    const double eh[3] = {d[0] / rho, d[1] / rho, d[2] / rho};
    const double sag = OMG / CLIGHT * (F.sat_vel[0] * Pe[1] + F.sat_pos[0] * Ve[1] - F.sat_vel[1] * Pe[0] - F.sat_pos[1] * Ve[0]);
    const double av[3] = {F.sat_vel[0] - Ve[0], F.sat_vel[1] - Ve[1], F.sat_vel[2] - Ve[2]};
    const double ae = d_dot3_nc(av, eh);
    out[0] = (ae + sag + ddt - F.sv_ddt + F.doppler * F.lamda) / F.var;
    double gP[3], gV[3];
    for (int k = 0; k < 3; ++k) { gP[k] = -(av[k] - ae * eh[k]) / rho; gV[k] = -eh[k]; }
    gP[0] += OMG / CLIGHT * (-F.sat_vel[1]); gP[1] += OMG / CLIGHT * F.sat_vel[0];
    gV[0] += OMG / CLIGHT * (-F.sat_pos[1]); gV[1] += OMG / CLIGHT * F.sat_pos[0];
    const double iv = 1.0 / F.var;
    for (int c = 0; c < 3; ++c) {
        const double gPl = gP[0] * Rf[c] + gP[1] * Rf[3 + c] + gP[2] * Rf[6 + c];
        const double gVl = gV[0] * Rf[c] + gV[1] * Rf[3 + c] + gV[2] * Rf[6 + c];
        out[1 + c] = F.ratio * gPl * iv; out[4 + c] = F.ratio * gVl * iv;
        out[7 + c] = (1.0 - F.ratio) * gPl * iv; out[10 + c] = (1.0 - F.ratio) * gVl * iv;
    }
    out[13] = iv;
}

// prior: blocks packed as prm[b*9 ..]; out: res[n] | J (n x 4 per block, packed at 4*n*b)
__global__ __launch_bounds__(256) void k_eval_marg(const double* __restrict__ J0, const double* __restrict__ r0, const double* __restrict__ x0, const int* __restrict__ kind,
                                                    const int* __restrict__ idx, int n, int nb, const double* __restrict__ prm, double* out) {
    extern __shared__ double dx[];           // [n] then per-block sign [nb] and Qleft rows [nb][12]
    double* sg = dx + n;
    double* Lq = sg + nb;
    const int tid = threadIdx.x;
    for (int b = tid; b < nb; b += blockDim.x) {
        const double* x = prm + 9 * b; const double* xb0 = x0 + 9 * b;
        if (kind[b] == GLIO_BLK_QUAT) {
            double q0inv[4], dq[4], L[16];
            d_qinv(xb0, q0inv);
            d_qmul(q0inv, x, dq);
            const double s = dq[0] >= 0 ? 2.0 : -2.0;                 // MarginalizationFactor.cpp:246-252, 276-281
            d_qnormalize(dq);
            for (int k = 0; k < 3; ++k) dx[idx[b] + k] = s * dq[1 + k];
            sg[b] = s;
            // Qleft(q0inv) rows 1..3
            const double w = q0inv[0], a = q0inv[1], bb = q0inv[2], c = q0inv[3];
            L[0] = w; L[1] = -a; L[2] = -bb; L[3] = -c;
            L[4] = a; L[5] = w; L[6] = -c; L[7] = bb;
            L[8] = bb; L[9] = c; L[10] = w; L[11] = -a;
            L[12] = c; L[13] = -bb; L[14] = a; L[15] = w;
            for (int k = 0; k < 12; ++k) Lq[12 * b + k] = L[4 + k];
        } else {
            const int size = kind[b] == GLIO_BLK_TRANS ? 3 : 9;
            for (int k = 0; k < size; ++k) dx[idx[b] + k] = x[k] - xb0[k];
            sg[b] = 0;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        double s = r0[i];
        for (int k = 0; k < n; ++k) s += J0[(size_t)i * n + k] * dx[k];
        out[i] = s;
    }
    for (int b = 0; b < nb; ++b) {
        double* M = out + n + (size_t)9 * n * b;                      // up to n x 9 per block
        if (kind[b] == GLIO_BLK_QUAT) {
            for (int e = tid; e < n * 4; e += blockDim.x) {
                const int i = e >> 2, c = e & 3;
                double a = 0;
                for (int k = 0; k < 3; ++k) a += J0[(size_t)i * n + idx[b] + k] * Lq[12 * b + k * 4 + c];
                M[e] = sg[b] * a;
            }
        } else {
            const int size = kind[b] == GLIO_BLK_TRANS ? 3 : 9;
            for (int e = tid; e < n * size; e += blockDim.x) M[e] = J0[(size_t)(e / size) * n + idx[b] + e % size];
        }
    }
}

__device__ void dqv_row(const double q[4], const double v[3], const double row[3], double acc[4]) {   // row . d(q v)/dq, Eigen _transformVector
    const double w = q[0]; const double* u = q + 1;
    double uv[3];
    d_cross(u, v, uv);
    acc[0] += 2 * d_dot3(row, uv);
    // columns: -2 w [v]x - 2 [u x v]x - 2 [u]x [v]x
    for (int k = 0; k < 3; ++k) {
        double ek[3] = {0, 0, 0}; ek[k] = 1.0;
        double vxe[3], uvxe[3], t[3], uxt[3];
        d_cross(v, ek, vxe);            // [v]x e_k
        d_cross(uv, ek, uvxe);          // [u x v]x e_k
        d_cross(v, ek, t); d_cross(u, t, uxt);     // [u]x [v]x e_k
        double col[3];
        for (int r = 0; r < 3; ++r) col[r] = -2 * w * vxe[r] - 2 * uvxe[r] - 2 * uxt[r];
        acc[1 + k] += d_dot3(row, col);
    }
}
// out: res | J_t1[3] | J_q1[4] | J_t2[3] | J_q2[4]
__global__ void k_eval_binary(const float4 cp, const double* __restrict__ prm /* pnc6 score t1 q1 t2 q2 */, double* out) {
    if (threadIdx.x != 0) return;
    const double* pnc = prm; const double score = prm[6];
    const double* t1 = prm + 7; const double* q1 = prm + 10; const double* t2 = prm + 14; const double* q2 = prm + 17;
    const double p[3] = {(double)cp.x, (double)cp.y, (double)cp.z};
    double pw[3], no[3], co[3];
    d_qrot(q1, p, pw);
    for (int k = 0; k < 3; ++k) pw[k] += t1[k];
    d_qrot(q2, pnc, no);
    d_qrot(q2, pnc + 3, co);
    for (int k = 0; k < 3; ++k) co[k] += t2[k];
    const double diff[3] = {pw[0] - co[0], pw[1] - co[1], pw[2] - co[2]};
    out[0] = score * d_dot3(no, diff);
    for (int k = 0; k < 3; ++k) { out[1 + k] = score * no[k]; out[8 + k] = -score * no[k]; }
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    const double r1[3] = {score * no[0], score * no[1], score * no[2]};
    dqv_row(q1, p, r1, a1);
    const double r2a[3] = {score * diff[0], score * diff[1], score * diff[2]}, r2b[3] = {-score * no[0], -score * no[1], -score * no[2]};
    dqv_row(q2, pnc, r2a, a2);
    dqv_row(q2, pnc + 3, r2b, a2);
    for (int k = 0; k < 4; ++k) { out[4 + k] = a1[k]; out[11 + k] = a2[k]; }
}

// scratch: one device blob [in | out], one host round trip
template <typename LaunchFn>
static int run_eval(glio_ctx* c, const void* in, size_t in_bytes, size_t out_doubles, std::vector<double>& out, LaunchFn launch) {
    EV_CHECK(hipSetDevice(c->device));
    unsigned char* d = nullptr;
    const size_t in_pad = (in_bytes + 15) & ~(size_t)15;
    EV_CHECK(hipMalloc((void**)&d, in_pad + out_doubles * 8));
    EV_CHECK(hipMemcpyAsync(d, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    launch(d, reinterpret_cast<double*>(d + in_pad));
    out.resize(out_doubles);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d + in_pad, out_doubles * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) { glio_set_error("evaluator launch failed: %s", hipGetErrorString(e)); return GLIO_E_HIP; }
    return GLIO_OK;
}

void glio_host_ecef_local(const double anc[3], double yaw, double R[9]);       // capi.hip: R_ecef_enu(anchor) * Rz(yaw)

extern "C" {

int glio_eval_dd_psr(glio_ctx* c, const glio_dd_psr* f, double const* const* P, double* res, double** J) {
    if (!c || !f || !P || !res) return GLIO_E_ARG;
    if (f->n_sat < 2 || f->n_sat > GLIO_DD_MAX_SAT || f->master < 0 || f->master >= f->n_sat) { glio_set_error("bad DD factor"); return GLIO_E_ARG; }
    struct { glio_dd_psr F; double prm[18]; } in;
    in.F = *f;
    memcpy(in.prm, P[0], 24); memcpy(in.prm + 3, P[1], 24);
    glio_host_ecef_local(P[3], P[2][0], in.prm + 6);                           // dd_psr_factor.hpp:33-45
    memcpy(in.prm + 15, P[3], 24);
    std::vector<double> out;
    const int rc = run_eval(c, &in, sizeof in, 19 + 114, out, [&](unsigned char* d, double* o) {
        hipLaunchKernelGGL(k_eval_dd, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<const glio_dd_psr*>(d),
                           reinterpret_cast<const double*>(d + offsetof(decltype(in), prm)), o);
    });
    if (rc) return rc;
    memcpy(res, out.data(), 19 * 8);
    if (J) { if (J[0]) memcpy(J[0], out.data() + 19, 57 * 8); if (J[1]) memcpy(J[1], out.data() + 19 + 57, 57 * 8); }
    return GLIO_OK;
}

int glio_eval_doppler(glio_ctx* c, const glio_doppler* f, double const* const* P, double* res, double** J) {
    if (!c || !f || !P || !res) return GLIO_E_ARG;
    struct { glio_doppler F; double prm[16]; } in;
    in.F = *f;
    memcpy(in.prm, P[0], 24); memcpy(in.prm + 3, P[1], 24); memcpy(in.prm + 6, P[2], 24); memcpy(in.prm + 9, P[3], 24);
    in.prm[12] = P[4][f->epoch];                                               // dopp_factor.hpp:38
    memcpy(in.prm + 13, P[6], 24);
    std::vector<double> out;
    const int rc = run_eval(c, &in, sizeof in, 14, out, [&](unsigned char* d, double* o) {
        hipLaunchKernelGGL(k_eval_doppler, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<const glio_doppler*>(d),
                           reinterpret_cast<const double*>(d + offsetof(decltype(in), prm)), o);
    });
    if (rc) return rc;
    res[0] = out[0];
    if (J) {
        if (J[0]) memcpy(J[0], &out[1], 24);
        if (J[1]) { memset(J[1], 0, 72); memcpy(J[1], &out[4], 24); }
        if (J[2]) memcpy(J[2], &out[7], 24);
        if (J[3]) { memset(J[3], 0, 72); memcpy(J[3], &out[10], 24); }
        if (J[4]) J[4][0] = out[13];
    }
    return GLIO_OK;
}

int glio_eval_marginalization(glio_ctx* c, const glio_prior* p, double const* const* P, double* res, double** J) {
    if (!c || !p || !P || !res || p->n <= 0 || p->n_blocks <= 0) return GLIO_E_ARG;
    const int n = p->n, nb = p->n_blocks;
    // blob: J0 [n*n] r0 [n] x0 [nb*9] prm [nb*9] | kind [nb] idx [nb] (ints)
    std::vector<double> blob((size_t)n * n + n + 18 * (size_t)nb + nb + 1);
    double* w = blob.data();
    memcpy(w, p->lin_jac, (size_t)n * n * 8); w += (size_t)n * n;
    memcpy(w, p->lin_res, n * 8); w += n;
    memcpy(w, p->blk_x0, (size_t)nb * 72); w += 9 * nb;
    for (int b = 0; b < nb; ++b) {
        const int size = p->blk_kind[b] == GLIO_BLK_TRANS ? 3 : (p->blk_kind[b] == GLIO_BLK_QUAT ? 4 : 9);
        memset(w + 9 * b, 0, 72); memcpy(w + 9 * b, P[b], size * 8);
    }
    w += 9 * nb;
    int* iw = reinterpret_cast<int*>(w);
    for (int b = 0; b < nb; ++b) { iw[b] = p->blk_kind[b]; iw[nb + b] = p->blk_idx[b]; }
    std::vector<double> out;
    const size_t lds = ((size_t)n + 13 * (size_t)nb) * 8;
    const int rc = run_eval(c, blob.data(), blob.size() * 8, (size_t)n + 9 * (size_t)n * nb, out, [&](unsigned char* d, double* o) {
        const double* dd = reinterpret_cast<const double*>(d);
        const double* dJ0 = dd; const double* dr0 = dJ0 + (size_t)n * n; const double* dx0 = dr0 + n; const double* dprm = dx0 + 9 * nb;
        const int* dk = reinterpret_cast<const int*>(dprm + 9 * nb);
        hipLaunchKernelGGL(k_eval_marg, dim3(1), dim3(256), lds, c->stream, dJ0, dr0, dx0, dk, dk + nb, n, nb, dprm, o);
    });
    if (rc) return rc;
    memcpy(res, out.data(), n * 8);
    if (J)
        for (int b = 0; b < nb; ++b) {
            if (!J[b]) continue;
            const int size = p->blk_kind[b] == GLIO_BLK_TRANS ? 3 : (p->blk_kind[b] == GLIO_BLK_QUAT ? 4 : 9);
            memcpy(J[b], out.data() + n + (size_t)9 * n * b, (size_t)n * size * 8);
        }
    return GLIO_OK;
}

int glio_eval_binary_plane(glio_ctx* c, const float cp[4], const double norm_cent[6], double score, double const* const* P, double* res, double** J) {
    if (!c || !cp || !norm_cent || !P || !res) return GLIO_E_ARG;
    double prm[21];
    memcpy(prm, norm_cent, 48); prm[6] = score;
    memcpy(prm + 7, P[0], 24); memcpy(prm + 10, P[1], 32); memcpy(prm + 14, P[2], 24); memcpy(prm + 17, P[3], 32);
    const float4 cpv = make_float4(cp[0], cp[1], cp[2], cp[3]);
    std::vector<double> out;
    const int rc = run_eval(c, prm, sizeof prm, 15, out, [&](unsigned char* d, double* o) {
        hipLaunchKernelGGL(k_eval_binary, dim3(1), dim3(64), 0, c->stream, cpv, reinterpret_cast<const double*>(d), o);
    });
    if (rc) return rc;
    res[0] = out[0];
    if (J) { if (J[0]) memcpy(J[0], &out[1], 24); if (J[1]) memcpy(J[1], &out[4], 32); if (J[2]) memcpy(J[2], &out[8], 24); if (J[3]) memcpy(J[3], &out[11], 32); }
    return GLIO_OK;
}

}  // extern "C"
