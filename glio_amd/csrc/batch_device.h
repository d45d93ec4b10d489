// batch_device.h -- the batch stage's context, shared by batch_kernels.hip (K8 linearisation, sequential banded solve, C-ABI),
// batch_solve_kernels.hip (block cyclic reduction) and batch_tr_kernels.hip (small factors, trust-region solve).
#pragma once
#include "glio_device.h"

#define BP_REC 56          // doubles per pair record (55 used)
#define BP_GRAM 45

struct glio_batch {
    int device;
    hipStream_t own_stream, stream;
    int K, band;
    int64_t max_con, n_con;
    float4* d_cp; double* d_nc; double* d_score;      // owned buffers (host upload path)
    const float4* cp; const double* nc; const double* score;   // active (owned or borrowed device pointers)
    int n_pairs, max_pairs;
    int* d_pair_i; int* d_pair_j; long long* d_pair_off;
    double* d_pair_rec;        // [max_pairs][BP_REC]
    int* d_pair_index;         // [K][2*band+1]
    double* d_poses;           // [K][7]
    double* d_M;               // [K][band+1][36] factor workspace (lower blocks (k+d, k))
    double* d_y;               // [K][6]
    double* d_delta;           // [K][6]
    double* d_newposes;        // [K][7]
    double* d_scalar;          // [4]
    double* d_parts;           // per-workgroup parts of the model decrease (summed in fixed order)
    double* h_poses; double* h_scalar;    // pinned
    hipEvent_t ev0, ev1;
    struct BatchSmall* small;  // delta_q / DD-pseudorange factors of the batch problem + trust-region workspaces (batch_tr_kernels.hip)
    void* bcr;                 // block-cyclic-reduction solver (batch_solve_kernels.hip); null for bands it does not cover
    int solver_mode;           // 1 = block cyclic reduction (default when available), 0 = the sequential banded kernels
    double* d_moments; int moments_pairs;      // per-pair moment records of K8 (batch_kernels.hip: k_batch_moments), sized for moments_pairs pairs
    int moments_valid;         // the records belong to the present constraint set (cleared by every glio_batch_set_constraints*)
    int n_mom_changed;         // moments_valid: pairs whose constraints were replaced since (their slots in d_mom_slots), taken again at the next solve
    int* d_mom_slots; int mom_slots_cap;
    int* h_prev_pi; int* h_prev_pj; int h_prev_n;      // the pair list the records were taken for
    long long* h_prev_off;                             // [2 h_prev_n] and each pair's record range then: a pair whose range moved counts as replaced
    int src_min, src_max;      // smallest / largest SOURCE keyframe (ci) of the present constraint set (src_max < src_min: empty); a sharded solve checks them against its range
};
// which of a pair of buffers a kernel of the device-resident batch solve works on, and whether it runs at all:
// buffer = cur ? (*cur ^ want) : want   (want 0: the current point's, 1: the candidate's);  *skip != 0: the kernel returns
struct BtSel { const int* cur; const int* skip; int want; };
__device__ __forceinline__ int bt_pick(const BtSel& s) { return s.cur ? (((*s.cur) ^ s.want) & 1) : s.want; }
__device__ __forceinline__ bool bt_skip(const BtSel& s) { return s.skip && *s.skip != 0; }

// ---- the normal matrix of the batch problem as an operator: pose band (K8 + small factors) + IMU chain records
struct HView { const double* Hg; const PairBlock* imu; int K, band, B; };
// unscaled entry (r, c) of block (ka, kb); keyframes beyond K are identity padding of the last super-block
__device__ __forceinline__ double h_entry(const HView& v, const int ka, const int r, const int kb, const int c) {
    if (ka >= v.K || kb >= v.K) return (ka == kb && r == c) ? 1.0 : 0.0;
    const int d = kb - ka, bw = v.band + 1;
    double s = 0.0;
    if (r < 6 && c < 6 && d <= v.band && d >= -v.band)
        s = d >= 0 ? v.Hg[((size_t)ka * bw + d) * 36 + r * 6 + c] : v.Hg[((size_t)kb * bw + (-d)) * 36 + c * 6 + r];
    if (v.imu) {
        if (d == 0) {
            if (ka < v.K - 1) s += v.imu[ka].H[r * 30 + c];
            if (ka > 0) s += v.imu[ka - 1].H[(15 + r) * 30 + 15 + c];
        } else if (d == 1) s += v.imu[ka].H[r * 30 + 15 + c];
        else if (d == -1) s += v.imu[kb].H[(15 + r) * 30 + c];
    }
    return s;
}

// ---- block cyclic reduction (batch_solve_kernels.hip)
// The matrix the solver factors: S (H_pose_band + H_imu_chain) S + shift, right-hand side S g -- read in place (no scaled copy).
struct BcrOp {
    const double* Hg[2];       // pose band buffers [K][band+1][36] | g [6 K] | cost of the current / candidate point
    const PairBlock* imu[2];   // IMU edge records (edge e between keyframes e, e+1) or null (pose-only problem)
    const double* gfull[2];    // full gradient [K B] (the all-reduced assembly buffer); null: g is read from Hg (B = 6)
    const int* cur;            // device pointer to the index of the current buffers (null: 0)
    const double* sc;          // Jacobi scale [K B] or null
    const double* dadd;        // additive diagonal [K B] (mu D^2) or null
    double lambda;             // when dadd == null: diag += lambda diag + 1e-12
    const int* skip;           // device flag: when *skip != 0 every kernel returns at once (null: never)
};
void* glio_bcr_create(int K, int band);
void* glio_bcr_create2(int K, int band, int B, int rank, int world);
void glio_bcr_owned_range(void* h, int* lo, int* hi);
double* glio_bcr_sepbuf(void* h, long long* count);           // the buffer the caller all-reduces between the two phases (last 16 doubles: the caller's)
int* glio_bcr_fail_flag(void* h);
void glio_bcr_enqueue_local(void* h, const BcrOp& op, hipStream_t stream);
void glio_bcr_enqueue_finish(void* h, const BcrOp& op, double* delta, hipStream_t stream);
int glio_bcr_levels(void* h);
void glio_bcr_destroy(void* h);
void glio_bcr_solve(void* h, const double* Hg, double lambda, double* delta, int** fail_dev, hipStream_t stream);
// the same with an explicit additive diagonal (dadd [6 K], may be null) instead of / on top of lambda diag(H): the trust-region
// solve passes mu D^2 of the Jacobi-scaled system here
void glio_bcr_solve_shift(void* h, const double* Hg, double lambda, const double* dadd, double* delta, int** fail_dev, hipStream_t stream);
// batch_kernels.hip: linearise this rank's shard into Hg_dev from the poses already on the device (b->d_poses)
void glio_batch_enqueue_linearize(glio_batch* b, double* Hg_dev);
// the same inside the device-resident solve: poses / output selected on the device (sel), rows [k0, k1) of the band only
// mode 0: stream the constraints (k_batch_pairs); 1: take the pairs' moments at the selected poses, then evaluate them (the first linearisation of
// a solve); 2: evaluate the stored moments only (every later one).  glio_batch_moments_ensure sizes the moment buffer for the present pairs.
void glio_batch_enqueue_linearize_sel(glio_batch* b, const BtSel& sel, const double* poses0, const double* poses1, double* Hg0, double* Hg1, int k0, int k1, int mode = 0);
int glio_batch_moments_ensure(glio_batch* b);
// factor_kernels.hip: ImuFactor of the batch chain, one workgroup per edge e in [e0, e1): rec[sel][e]
void glio_launch_batch_imu(hipStream_t stream, const BtSel& sel, double gravity, const ImuEdgeDev* edges, int e0, int e1, const double* poses0, const double* poses1,
                           const double* sb0, const double* sb1, PairBlock* rec0, PairBlock* rec1);
// capi.hip: pre-digest of one pre-integration (sqrt_info = LLT(cov^-1).L^T, bias Jacobian blocks)
bool glio_digest_imu_edge(const glio_preint* p, int slot, ImuEdgeDev* e);
// batch_tr_kernels.hip
void glio_batch_small_destroy(glio_batch* b);
// capi.hip: GLIO_DEBUG_LDS_POISON=1 fills the LDS of every CU with NaNs (a kernel that reads LDS it never wrote then fails loudly); no-op otherwise
void glio_lds_poison_stream(hipStream_t stream);
