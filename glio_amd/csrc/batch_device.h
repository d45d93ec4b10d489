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
};
void* glio_bcr_create(int K, int band);
void glio_bcr_destroy(void* h);
void glio_bcr_solve(void* h, const double* Hg, double lambda, double* delta, int** fail_dev, hipStream_t stream);
// the same with an explicit additive diagonal (dadd [6 K], may be null) instead of / on top of lambda diag(H): the trust-region
// solve passes mu D^2 of the Jacobi-scaled system here
void glio_bcr_solve_shift(void* h, const double* Hg, double lambda, const double* dadd, double* delta, int** fail_dev, hipStream_t stream);
// batch_kernels.hip: linearise this rank's shard into Hg_dev from the poses already on the device (b->d_poses)
void glio_batch_enqueue_linearize(glio_batch* b, double* Hg_dev);
// batch_tr_kernels.hip
void glio_batch_small_destroy(glio_batch* b);

