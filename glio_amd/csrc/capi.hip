// capi.hip -- the extern "C" boundary of libglio_hip.so (declared in include/glio_hip.h): context
// management, uploads (with the host-side digestion the reference also does on the CPU before the hot
// path: IMU sqrt-information, factor sorting), and the solve / linearise / evaluator orchestration.
// There is NO CPU fallback in here: without a HIP device glio_create fails and every entry point
// returns GLIO_E_HIP / GLIO_E_STATE.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "glio_device.h"

void glio_launch_eval_imu(glio_ctx* c, const ImuEdgeDev* d_edge, const double* d_params, double* d_out);
void glio_launch_eval_lidar(glio_ctx* c, const float cp[4], const float plane[4], double score, const double* d_params, double* d_out);
void glio_launch_gram(glio_ctx* c, int np);
int glio_tr_step_configure(size_t max_lds);

static thread_local char g_err[512] = "";
void glio_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct CtxExtra {
    GnssDevExtra gx;
    double* d_eval_params; double* d_eval_out; ImuEdgeDev* d_eval_edge;
    double* h_eval;   // pinned
    int imu_edge0;    // index of the IMU edge leaving slot 0 (-1: none)
    // the pre-integrations of the last glio_set_imu and what was derived from them (inverse covariance root, 15^3 flops each): after a
    // slide W - 2 of the W - 1 edges are the same pre-integrations one slot lower, and only the new one is digested again
    std::vector<glio_preint> imu_raw; std::vector<ImuEdgeDev> imu_dig;
    // early uploads (glio_set_imu, glio_set_gnss): a stream of their own and two pinned blocks with device mirrors, see stage_begin_early
    hipEvent_t ev_copy = nullptr;          // glio_set_scan: the end of the scan's copy (what the call waits for; the presort behind it is not waited for)
    hipStream_t up_stream = nullptr; hipEvent_t ev_up = nullptr;
    struct UpArena { char* h = nullptr; char* d = nullptr; size_t cap = 0; hipEvent_t ev_free = nullptr; bool pending = false; hipEvent_t ev_copied = nullptr; bool copying = false; } up[2];
    int up_next = 0, up_cur = -1, up_mode = -1, up_wait = -1, unstage_up = -1;
    int* marg_h_ok = nullptr;              // glio_marginalize_keep_async: the pinned "positive definite" flag its finish looks at
    hipEvent_t ev_ahead = nullptr; int ahead_valid = 0, ahead_n = 0;      // glio_set_scan_ahead: the upload + presort of the NEXT keyframe's scan on the upload stream
    char* sv_h = nullptr; char* sv_d = nullptr; size_t sv_cap = 0;
};
// the extras hang off the context itself (glio_ctx::extra): no process-global registry, so independent contexts can be
// created, used and destroyed from different threads concurrently (Estimator.cpp:5398-5404)
static CtxExtra* extra_of(glio_ctx* c) { return static_cast<CtxExtra*>(c->extra); }
GnssDevExtra* glio_extra(glio_ctx* c) { return &extra_of(c)->gx; }
static int marg_pending_done(glio_ctx* c);

// ---- roctx ranges (GLIO_ROCTX=1)
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("GLIO_ROCTX");
        if (!e || atoi(e) == 0) return;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
GlioTraceRange::GlioTraceRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
GlioTraceRange::~GlioTraceRange() { if (on) roctx().pop(); }

extern "C" {

int glio_abi_version(void) { return 4; }   // 3: glio_opts.lidar_precision, the batch pose problem (small factors, trust-region solve), 9 struct sizes
const char* glio_last_error(void) { return g_err; }
int glio_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int glio_struct_sizes(int32_t* out, int n) {
    const int32_t v[9] = {(int32_t)sizeof(glio_opts), (int32_t)sizeof(glio_state), (int32_t)sizeof(glio_preint), (int32_t)sizeof(glio_prior),
                          (int32_t)sizeof(glio_dd_psr), (int32_t)sizeof(glio_doppler), (int32_t)sizeof(glio_gnss_frame), (int32_t)sizeof(glio_summary),
                          (int32_t)sizeof(glio_batch_tr_opts)};
    for (int i = 0; i < n && i < 9; ++i) out[i] = v[i];
    return 9;
}

void glio_opts_default(glio_opts* o) {
    memset(o, 0, sizeof *o);
    o->window = 5; o->max_iterations = 15; o->max_points_per_scan = 65536; o->max_map_points = 1 << 21;
    o->max_ddt_epochs = 0; o->jacobi_scaling = 1;
    o->huber_delta = 1.0; o->doppler_huber_delta = 1.0;
    o->q_lb[0] = 1.0; o->t_lb[2] = 0.28;
    o->lidar_const = 7.5; o->surf_dist_thres = 0.18; o->kd_max_radius = 1.5; o->weight_gate = 0.3; o->trust_region_strategy = GLIO_STRATEGY_DOGLEG; o->unit_scores = 0;
    o->gravity = 9.80511;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
}

static int g_fill = getenv("GLIO_DEBUG_FILL") ? atoi(getenv("GLIO_DEBUG_FILL")) : -1;     // development aid: fill every allocation with this byte
#define ALLOC(ptr, bytes) do { GLIO_HIP_CHECK(hipMalloc((void**)&(ptr), (bytes) > 0 ? (size_t)(bytes) : 16)); \
                               if (g_fill >= 0) GLIO_HIP_CHECK(hipMemset((void*)(ptr), g_fill, (bytes) > 0 ? (size_t)(bytes) : 16)); } while (0)

static int create_body(int device, const glio_opts* opts, glio_ctx* c);
int glio_create(int device, const glio_opts* opts, glio_ctx** out) {
    if (!opts || !out) { glio_set_error("null argument"); return GLIO_E_ARG; }
    if (glio_device_count() < 1) { glio_set_error("no HIP device visible: the GLIO hot path has no CPU fallback"); return GLIO_E_HIP; }
    const int W = opts->window;
    if (W < 1 || W > GLIO_MAX_WINDOW || opts->max_points_per_scan < 1) { glio_set_error("bad window / capacity"); return GLIO_E_ARG; }
    const int n_max = 15 * W + std::max(0, opts->max_ddt_epochs);
    if (glio_tr_step_lds_bytes(n_max) > 160 * 1024) { glio_set_error("15*W + ddt = %d unknowns exceed the single-CU solver (LDS)", n_max); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(device));
    glio_ctx* c = new glio_ctx();
    memset(c, 0, sizeof *c);
    c->device = device;
    const int rc = create_body(device, opts, c);
    if (rc != GLIO_OK) {                      // a failed allocation half way: release what was built (every pointer is null-checked)
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        glio_destroy(c);
        memcpy(g_err, keep, sizeof keep);
        return rc;
    }
    *out = c;
    return GLIO_OK;
}
static int create_body(int device, const glio_opts* opts, glio_ctx* c) {
    const int W = opts->window;
    const int n_max = 15 * W + std::max(0, opts->max_ddt_epochs);
    c->opts = *opts; c->device = device; c->W = W; c->cap = opts->max_points_per_scan;
    c->n_ddt_max = std::max(0, opts->max_ddt_epochs); c->n_max = n_max;
    {   // the window's kernels are short and latency-bound (one-CU trust-region steps, marginalization, local map): highest priority, so that wide launches of
        // other streams (the batch association of the same keyframe) do not sit in front of them
        int least = 0, greatest = 0;
        const char* e = getenv("GLIO_CTX_PRIORITY");
        if ((!e || atoi(e) != 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            GLIO_HIP_CHECK(hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, greatest));
        else GLIO_HIP_CHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    }
    c->stream = c->own_stream;
    { hipDeviceProp_t prop; c->n_cu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 0; }
    const size_t wc = (size_t)W * c->cap;
    ALLOC(c->d_pts, wc * sizeof(float4)); ALLOC(c->d_planes, wc * sizeof(float4)); ALLOC(c->d_scores, wc * sizeof(double));
    ALLOC(c->d_count, W * sizeof(int)); ALLOC(c->d_scan, wc * sizeof(float4));
    if (opts->lidar_precision == GLIO_LIDAR_F32_MFMA) { ALLOC(c->d_pts_s, wc * sizeof(float4)); c->f32_dirty = 1; }
    else if (opts->lidar_precision != GLIO_LIDAR_F64) { glio_set_error("unknown lidar_precision %d", opts->lidar_precision); return GLIO_E_ARG; }
    // NB: every memset goes on the context's (non-blocking) stream: a null-stream hipMemset is asynchronous
    // and unordered with it, and once raced with the first glio_set_correspondences count upload.
    GLIO_HIP_CHECK(hipMemsetAsync(c->d_count, 0, W * sizeof(int), c->stream));
    ALLOC(c->d_imu, W * sizeof(ImuEdgeDev)); ALLOC(c->d_imu_blocks, 2 * W * sizeof(PairBlock));
    ALLOC(c->d_gnss_blocks, 2 * (size_t)W * W * sizeof(PairBlock));
    ALLOC(c->d_groups, (size_t)W * W * sizeof(GnssGroup));
    const int ne = std::max(1, c->n_ddt_max);
    ALLOC(c->d_ddt_blocks, 2 * (size_t)ne * sizeof(DdtBlock));
    GLIO_HIP_CHECK(hipMemsetAsync(c->d_ddt_blocks, 0, 2 * (size_t)ne * sizeof(DdtBlock), c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int npmax = 6 * W + 9;
    ALLOC(c->d_prior_J0, (size_t)npmax * npmax * 8); ALLOC(c->d_prior_A0, (size_t)npmax * npmax * 8);
    ALLOC(c->d_prior_r0, npmax * 8); ALLOC(c->d_prior_x0, (size_t)(2 * W + 1) * 9 * 8);
    ALLOC(c->d_prior_slot, (2 * W + 1) * 4); ALLOC(c->d_prior_kind, (2 * W + 1) * 4); ALLOC(c->d_prior_idx, (2 * W + 1) * 4);
    ALLOC(c->d_prior_index, 15 * W * 4);
    ALLOC(c->d_prior_H, 2 * (size_t)npmax * npmax * 8); ALLOC(c->d_prior_g, 2 * npmax * 8); ALLOC(c->d_prior_cost, 2 * 8);
    ALLOC(c->d_prior_work, (size_t)(3 * npmax + 9 * (2 * W + 1)) * 8);
    const int nx = glio_x_size(W, c->n_ddt_max);
    {   // [SolverStatus | pad to 512 B | x buffer 0] in one allocation (and one pinned mirror): a solve starts with ONE upload
        unsigned char* up = nullptr;
        ALLOC(up, 512 + (size_t)nx * 8);
        c->d_status = reinterpret_cast<SolverStatus*>(up);
        c->d_x[0] = reinterpret_cast<double*>(up + 512);
    }
    for (int k = 0; k < 2; ++k) {
        if (k == 1) ALLOC(c->d_x[k], nx * 8);
        ALLOC(c->d_H[k], (size_t)n_max * n_max * 8);
        ALLOC(c->d_g[k], n_max * 8);
        ALLOC(c->d_cost[k], 8);
    }
    ALLOC(c->d_xout, nx * 8);
    ALLOC(c->d_lidar_partials, 2 * (size_t)W * GLIO_K3_MAX_BLOCKS_PER_KF * GLIO_LIDAR_ACC * 8);
    ALLOC(c->d_hdiag[0], n_max * 8); ALLOC(c->d_hdiag[1], n_max * 8);
    ALLOC(c->d_chain_tabs, (size_t)(8 + 15) * W * sizeof(short));
    ALLOC(c->d_chain_src, 2 * (size_t)W * GLIO_CS_SOURCES * GLIO_CS_STRIDE * 8);
    GLIO_HIP_CHECK(hipMemsetAsync(c->d_chain_src, 0, 2 * (size_t)W * GLIO_CS_SOURCES * GLIO_CS_STRIDE * 8, c->stream));
    GLIO_HIP_CHECK(hipHostMalloc((void**)&c->h_chain_tabs, 2 * (size_t)(8 + 15) * W * sizeof(short)));      // two copies, used in turn (glio_chain_tabs_upload)
    c->h_groups = (GnssGroup*)calloc((size_t)W * W, sizeof(GnssGroup));
    c->h_prior_index = (int*)malloc((size_t)15 * W * sizeof(int));
    for (int k = 0; k < 15 * W; ++k) c->h_prior_index[k] = -1;
    c->chain_tabs_dirty = 1; c->h_band_clean = 0;
    c->k3_bpk = GLIO_K3_BLOCKS_PER_KF; c->last_k3_nb = c->k3_bpk; c->merged_linearize = 2; c->want_pair_H = 1; c->k3_unroll = 22;   /* 2-deep batches, non-temporal loads, next batch issued before the arithmetic of the current one */
    { int per = 768 / W;   /* ~3 workgroups per CU measured best on MI355X (scripts/k3_sweep.py) */
      // the fp32 / MFMA form keeps two chunks (four 16-byte loads per lane) in flight per wavefront and wants ~8x as many of them resident: at the C5
      // shape 120 workgroups per keyframe measured best (8 / 15 / 30 / 60 / 120 / 240: 96.7 / 91.8 / 88.8 / 82.1 / 76.8 / 80.2 us, scripts/c5_launch.py --sweep)
      if (opts->lidar_precision == GLIO_LIDAR_F32_MFMA) per = 6000 / W;
      if (per < 8) per = 8; if (per > GLIO_K3_MAX_BLOCKS_PER_KF) per = GLIO_K3_MAX_BLOCKS_PER_KF; c->k3_bpk = per; }
    ALLOC(c->d_lidar_blocks, 2 * (size_t)W * GLIO_LIDAR_ACC * 8);
    ALLOC(c->d_L, (size_t)(n_max + 1) * n_max * 8);
    c->vstride = (n_max + 15) & ~15;
    ALLOC(c->d_vec, (size_t)10 * c->vstride * 8);
    {   // structured solver
        ArrowDev& ar = c->arrow;
        ar.mode = 1; ar.gnss_ok = 1; ar.prior_ok = 1; ar.max_epoch = -1; ar.gnss_chain = 1; ar.prior_chain = 1;
        ALLOC(ar.d_ep_slots, (size_t)ne * sizeof(int2)); ALLOC(ar.d_ep_off, (W + 1) * 4); ALLOC(ar.d_ep_list, 2 * (size_t)ne * 4);
        GLIO_HIP_CHECK(hipMemsetAsync(ar.d_ep_slots, 0xff, (size_t)ne * sizeof(int2), c->stream));
        GLIO_HIP_CHECK(hipMemsetAsync(ar.d_ep_off, 0, (W + 1) * 4, c->stream));
        ALLOC(ar.d_Y, (size_t)(c->n_ddt_max + 9 * W) * (6 * W + 2) * 8);
        ALLOC(ar.d_blk, (size_t)W * 544 * 8); ALLOC(ar.d_Lblk, (size_t)W * 190 * 8); ALLOC(ar.d_Sp, (size_t)(6 * W + 1) * 6 * W * 8);
        ALLOC(ar.d_z, (size_t)n_max * 8); ALLOC(ar.d_flag, 4); ALLOC(ar.d_dbg, 320 * 8);
        ALLOC(ar.d_chain_sum, (size_t)W * GLIO_CS_STRIDE * 8); ALLOC(ar.d_chain_done, (size_t)W * 4); ar.chain_seq = 1;
        ALLOC(ar.d_fat_ep, (size_t)ne * 34 * 8);
        GLIO_HIP_CHECK(hipMemsetAsync(ar.d_chain_done, 0, (size_t)W * 4, c->stream));
        GLIO_HIP_CHECK(hipMemsetAsync(ar.d_flag, 0, 4, c->stream));
        GLIO_HIP_CHECK(hipMemsetAsync(ar.d_dbg, 0, 320 * 8, c->stream));          // (slot 300 counts the steps that took the fat helpers' products)
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    GLIO_HIP_CHECK(hipHostMalloc((void**)&c->h_progress, 64, hipHostMallocMapped | hipHostMallocCoherent));
    GLIO_HIP_CHECK(hipHostGetDevicePointer((void**)&c->d_progress, (void*)c->h_progress, 0));
    c->h_progress[0] = 0; c->h_progress[1] = 0; c->h_progress[2] = 0; c->enqueue_lead = 1;
    static_assert(sizeof(SolverStatus) <= 512, "result layout");
    GLIO_HIP_CHECK(hipHostMalloc((void**)&c->h_result, 512 + (size_t)nx * 8, hipHostMallocMapped | hipHostMallocCoherent));
    GLIO_HIP_CHECK(hipHostGetDevicePointer((void**)&c->d_result, (void*)c->h_result, 0));
    memset(c->h_result, 0, 512 + (size_t)nx * 8);
    c->solve_id = 0;
    {
        unsigned char* hup = nullptr;
        GLIO_HIP_CHECK(hipHostMalloc((void**)&hup, 512 + (size_t)nx * 8, hipHostMallocMapped));
        GLIO_HIP_CHECK(hipHostGetDevicePointer(&c->d_h_status, (void*)hup, 0));
        c->h_status = reinterpret_cast<SolverStatus*>(hup);
        c->h_xbuf = reinterpret_cast<double*>(hup + 512);
    }
    GLIO_HIP_CHECK(hipEventCreate(&c->ev0)); GLIO_HIP_CHECK(hipEventCreate(&c->ev1));
    CtxExtra* ex = new CtxExtra();          // value-initialised: pointers null, vectors empty
    ex->imu_edge0 = -1;
    ALLOC(ex->gx.d_runs, (size_t)std::max(1, c->n_ddt_max) * sizeof(DopRun));
    ALLOC(ex->gx.d_prior_colblk, npmax * 4);
    ALLOC(ex->d_eval_params, 64 * 8); ALLOC(ex->d_eval_out, (15 + 15 * 32) * 8); ALLOC(ex->d_eval_edge, sizeof(ImuEdgeDev));
    GLIO_HIP_CHECK(hipHostMalloc((void**)&ex->h_eval, (15 + 15 * 32) * 8));
    c->extra = ex;
    if (glio_tr_step_configure(160 * 1024) != GLIO_OK) return GLIO_E_HIP;
    if (glio_assoc_create(c) != GLIO_OK) return GLIO_E_HIP;
    return GLIO_OK;
}

void glio_destroy(glio_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->ev_ext_read) { if (c->ext_read_pending) hipEventSynchronize(c->ev_ext_read); hipEventDestroy(c->ev_ext_read); c->ev_ext_read = nullptr; }
    if (c->extra && extra_of(c)->up_stream) hipStreamSynchronize(extra_of(c)->up_stream);       // (a scan sent ahead may still be on its way)
    if (c->extra && extra_of(c)->ev_ahead) { hipEventDestroy(extra_of(c)->ev_ahead); extra_of(c)->ev_ahead = nullptr; }
    glio_assoc_destroy(c);
    glio_localmap_destroy(c);
    void* ptrs[] = {c->d_pts, c->d_planes, c->d_scores, c->d_pts_s, c->d_count, c->d_scan, c->d_imu, c->d_imu_blocks, c->d_gnss_blocks, c->d_groups,
                    c->d_ddt_blocks, c->d_dd, c->d_dop, c->d_prior_J0, c->d_prior_A0, c->d_prior_r0, c->d_prior_x0, c->d_prior_slot,
                    c->d_prior_kind, c->d_prior_idx, c->d_prior_index, c->d_prior_H, c->d_prior_g, c->d_prior_cost, c->d_prior_work,
                    /* d_x[0] lives inside d_status' allocation */ c->d_x[1], c->d_H[0], c->d_H[1], c->d_g[0], c->d_g[1], c->d_cost[0], c->d_cost[1], c->d_xout,
                    c->d_lidar_partials, c->d_hdiag[0], c->d_hdiag[1], c->d_chain_tabs, c->d_chain_src, c->d_lidar_blocks, c->d_L, c->d_vec, c->d_status,
                    c->arrow.d_ep_slots, c->arrow.d_ep_off, c->arrow.d_ep_list, c->arrow.d_Y, c->arrow.d_blk, c->arrow.d_Lblk, c->arrow.d_Sp, c->arrow.d_z, c->arrow.d_flag, c->arrow.d_dbg, c->arrow.d_chain_sum, c->arrow.d_chain_done, c->arrow.d_fat_ep};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_status) hipHostFree(c->h_status); /* h_xbuf lives inside it */
    if (c->h_progress) hipHostFree((void*)c->h_progress);
    if (c->h_result) hipHostFree(c->h_result);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->raw_stage.d) { hipFree(c->raw_stage.d); c->raw_stage.d = nullptr; c->raw_stage.cap = 0; }
    if (c->h_chain_tabs) hipHostFree(c->h_chain_tabs);
    for (hipEvent_t& e : c->ev_tabs) if (e) { hipEventDestroy(e); e = nullptr; }
    free(c->h_groups); free(c->h_prior_index);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (CtxExtra* ex = extra_of(c)) {
        if (ex->ev_copy) hipEventDestroy(ex->ev_copy);
        if (ex->up_stream) hipStreamSynchronize(ex->up_stream);
        for (auto& a : ex->up) { if (a.h) hipHostFree(a.h); if (a.d) hipFree(a.d); if (a.ev_free) hipEventDestroy(a.ev_free); if (a.ev_copied) hipEventDestroy(a.ev_copied); }
        if (ex->ev_up) hipEventDestroy(ex->ev_up);
        if (ex->up_stream) hipStreamDestroy(ex->up_stream);
        if (ex->gx.d_runs) hipFree(ex->gx.d_runs);
        if (ex->gx.d_prior_colblk) hipFree(ex->gx.d_prior_colblk);
        if (ex->d_eval_params) hipFree(ex->d_eval_params);
        if (ex->d_eval_out) hipFree(ex->d_eval_out);
        if (ex->d_eval_edge) hipFree(ex->d_eval_edge);
        if (ex->h_eval) hipHostFree(ex->h_eval);
        delete ex;
        c->extra = nullptr;
    }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

int glio_set_stream(glio_ctx* c, void* s) {
    if (!c) return GLIO_E_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return GLIO_OK;
}
int glio_synchronize(glio_ctx* c) {
    if (!c) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- LiDAR
int glio_set_correspondences(glio_ctx* c, int slot, const float* pts, const float* planes, const double* scores, int n) {
    if (!c || slot < 0 || slot >= c->W || n < 0 || n > c->cap) { glio_set_error("bad slot / count %d (cap %d)", n, c ? c->cap : 0); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    const size_t off = (size_t)slot * c->cap;
    if (n > 0) {
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_pts + off, pts, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_planes + off, planes, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_scores + off, scores, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    }
    c->h_count[slot] = n;
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_count + slot, &c->h_count[slot], 4, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_factors = 1; c->f32_dirty = 1;
    return GLIO_OK;
}

int glio_get_correspondences(glio_ctx* c, int slot, float* pts, float* planes, double* scores, int capacity, int* out_count) {
    if (!c || slot < 0 || slot >= c->W) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    const int n = c->h_count[slot];
    if (out_count) *out_count = n;
    if (n > capacity) { glio_set_error("capacity %d < count %d", capacity, n); return GLIO_E_ARG; }
    const size_t off = (size_t)slot * c->cap;
    if (n > 0) {
        if (pts) GLIO_HIP_CHECK(hipMemcpyAsync(pts, c->d_pts + off, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
        if (planes) GLIO_HIP_CHECK(hipMemcpyAsync(planes, c->d_planes + off, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
        if (scores) GLIO_HIP_CHECK(hipMemcpyAsync(scores, c->d_scores + off, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    }
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GLIO_OK;
}

// ---- strided point input: the clouds as the caller holds them (pcl::PointCloud<pcl::PointXYZI>::points.data(): 32 B records, intensity at byte 16,
// GLIO/include/utils/common.h: PointType) -- one upload of the raw records + an unpack kernel, no host-side packing pass (Estimator.cpp:3529-3631 hands such
// clouds around)
__global__ void k_unpack_points(const unsigned char* __restrict__ raw, const int n, const int stride, const int ioff, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* r = raw + (size_t)i * stride;
    const float* p = reinterpret_cast<const float*>(r);
    out[i] = make_float4(p[0], p[1], p[2], *reinterpret_cast<const float*>(r + ioff));
}
int glio_point_layout_ok(int stride, int ioff) { return stride >= 16 && (stride & 3) == 0 && (ioff & 3) == 0 && ioff >= 12 && ioff + 4 <= stride; }
int glio_upload_points(hipStream_t stream, GlioRawStage* st, const void* host, int n, int stride, int ioff, float4* d_out) {
    if (n <= 0) return GLIO_OK;
    if (stride == 16 && ioff == 12) { GLIO_HIP_CHECK(hipMemcpyAsync(d_out, host, (size_t)n * 16, hipMemcpyHostToDevice, stream)); return GLIO_OK; }
    const size_t bytes = (size_t)n * stride;
    if (bytes > st->cap) {
        GLIO_HIP_CHECK(hipStreamSynchronize(stream));               // (an earlier unpack may still read the old buffer)
        if (st->d) hipFree(st->d);
        st->d = nullptr; st->cap = 0;
        GLIO_HIP_CHECK(hipMalloc(&st->d, bytes + bytes / 4));
        st->cap = bytes + bytes / 4;
    }
    GLIO_HIP_CHECK(hipMemcpyAsync(st->d, host, bytes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_unpack_points, dim3((n + 255) / 256), dim3(256), 0, stream, static_cast<const unsigned char*>(st->d), n, stride, ioff, d_out);
    return GLIO_OK;
}
int glio_set_map_strided(glio_ctx* c, const void* map_points, int n, int stride_bytes, int intensity_offset) {
    GLIO_TRACE("K1 glio_set_map (voxel hash build)");
    if (!c || !map_points || n < 0 || n > c->opts.max_map_points) { glio_set_error("bad map size"); return GLIO_E_ARG; }
    if (!glio_point_layout_ok(stride_bytes, intensity_offset)) { glio_set_error("bad point layout (stride %d, intensity at %d)", stride_bytes, intensity_offset); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    return glio_assoc_build_map(c, map_points, n, stride_bytes, intensity_offset);
}
int glio_set_map(glio_ctx* c, const float* map_xyzi, int n) { return glio_set_map_strided(c, map_xyzi, n, 16, 12); }
int glio_set_scan(glio_ctx* c, int slot, const float* scan, int n) { return glio_set_scan_strided(c, slot, scan, n, 16, 12); }
int glio_set_scan_strided(glio_ctx* c, int slot, const void* scan, int n, int stride_bytes, int intensity_offset) {
    if (!c || slot < 0 || slot >= c->W || n < 0 || n > c->cap || (n > 0 && !scan)) { glio_set_error("bad slot / scan size"); return GLIO_E_ARG; }
    if (!glio_point_layout_ok(stride_bytes, intensity_offset)) { glio_set_error("bad point layout (stride %d, intensity at %d)", stride_bytes, intensity_offset); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    // (a copy of a resident scan on another stream -- glio_bassoc_set_frame_from_scan -- may still be reading the row this call overwrites)
    if (c->ext_read_pending) { GLIO_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ext_read, 0)); c->ext_read_pending = 0; }
    { const int ru = glio_upload_points(c->stream, &c->raw_stage, scan, n, stride_bytes, intensity_offset, c->d_scan + (size_t)glio_scan_row(c, slot) * c->cap); if (ru != GLIO_OK) return ru; }
    // the caller's buffer must have been read when the call returns: that is the copy, not the presort enqueued behind it (0.03 ms that the next call's
    // launches now overlap)
    CtxExtra* ex = extra_of(c);
    if (!ex->ev_copy) GLIO_HIP_CHECK(hipEventCreateWithFlags(&ex->ev_copy, hipEventDisableTiming));
    GLIO_HIP_CHECK(hipEventRecord(ex->ev_copy, c->stream));
    glio_assoc_scan_uploaded(c, slot, n);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipEventSynchronize(ex->ev_copy));
    c->h_scan_count[slot] = n;
    return GLIO_OK;
}
// The NEXT keyframe's scan, sent while this keyframe's call is still running: into the ring row that becomes slot W - 1 with the next glio_slide_window -- the row of
// the current slot 0, whose scan nothing reads any more once the window's association is through (the call fetches its counts first) -- on the upload stream, the
// presort behind it there.  The next call's glio_slide_window takes it over (count, an event the context's stream waits for) and needs no glio_set_scan.
int glio_set_scan_ahead(glio_ctx* c, const float* scan, int n) { return glio_set_scan_ahead_strided(c, scan, n, 16, 12); }
int glio_set_scan_ahead_strided(glio_ctx* c, const void* scan, int n, int stride_bytes, int intensity_offset) {
    if (!c || c->W < 2 || n < 0 || n > c->cap || (n > 0 && !scan)) { glio_set_error("bad scan size"); return GLIO_E_ARG; }
    if (!glio_point_layout_ok(stride_bytes, intensity_offset)) { glio_set_error("bad point layout (stride %d, intensity at %d)", stride_bytes, intensity_offset); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }              // every search that reads slot 0's scan has ended
    CtxExtra* ex = extra_of(c);
    if (!ex->up_stream) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) GLIO_HIP_CHECK(hipStreamCreateWithPriority(&ex->up_stream, hipStreamNonBlocking, greatest));
        else GLIO_HIP_CHECK(hipStreamCreateWithFlags(&ex->up_stream, hipStreamNonBlocking));
        GLIO_HIP_CHECK(hipEventCreateWithFlags(&ex->ev_up, hipEventDisableTiming));
    }
    if (!ex->ev_ahead) GLIO_HIP_CHECK(hipEventCreateWithFlags(&ex->ev_ahead, hipEventDisableTiming));
    if (!ex->ev_copy) GLIO_HIP_CHECK(hipEventCreateWithFlags(&ex->ev_copy, hipEventDisableTiming));
    // (another object's stream may still be reading a resident scan -- never slot 0's: glio_bassoc_set_frame_from_scan copies the newest -- but the event is cheap)
    if (c->ext_read_pending) { GLIO_HIP_CHECK(hipStreamWaitEvent(ex->up_stream, c->ev_ext_read, 0)); }
    const size_t row = (size_t)glio_scan_row(c, 0) * c->cap;
    { const int ru = glio_upload_points(ex->up_stream, &c->raw_stage, scan, n, stride_bytes, intensity_offset, c->d_scan + row); if (ru != GLIO_OK) return ru; }
    GLIO_HIP_CHECK(hipEventRecord(ex->ev_copy, ex->up_stream));
    glio_assoc_presort_row(c, ex->up_stream, row, n);
    GLIO_HIP_CHECK(hipEventRecord(ex->ev_ahead, ex->up_stream));
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipEventSynchronize(ex->ev_copy));       // the caller's buffer has been read
    ex->ahead_valid = 1; ex->ahead_n = n;
    c->h_scan_count[0] = 0;                                  // (slot 0's scan is gone)
    return GLIO_OK;
}
// ... and the local map of the next keyframe's call, built during this one's tail too: the cloud glio_set_scan_ahead has just sent, pushed at the pose the caller
// has for the new keyframe (its initial pose follows from this call's solve and the odometry: known once the solve has returned), voxel grid, K1 -- all on the
// upload stream, beside the marginalization and the batch association (none of them reads the map).  The next call then makes no glio_localmap_push_scan /
// glio_localmap_build: its glio_slide_window waits for the event and the association finds the map.
int glio_localmap_push_scan_ahead_and_build(glio_ctx* c, const float lidar_offset[3], const double q[4], const double t[3], int* out_points) {
    if (!c) return GLIO_E_ARG;
    CtxExtra* ex = extra_of(c);
    if (!ex->ahead_valid || !ex->up_stream) { glio_set_error("glio_set_scan_ahead first"); return GLIO_E_STATE; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    // the local map's and K1's launches go where c->stream points: for the length of this call that is the upload stream; slot 0's row holds the new scan
    hipStream_t own = c->stream;
    const int n0 = c->h_scan_count[0];
    c->stream = ex->up_stream; c->h_scan_count[0] = ex->ahead_n;
    int rc = glio_localmap_push_scan(c, 0, lidar_offset, q, t);
    if (rc == GLIO_OK) rc = glio_localmap_build(c, out_points);
    hipError_t e = hipEventRecord(ex->ev_ahead, ex->up_stream);          // (replaces the scan's own event: the next slide waits for scan, presort, map and K1)
    c->stream = own; c->h_scan_count[0] = n0;
    if (rc != GLIO_OK) return rc;
    GLIO_HIP_CHECK(e);
    return GLIO_OK;
}
int glio_associate_resident(glio_ctx* c, int slot, const double q[4], const double t[3], int* out_count) {
    GLIO_TRACE("K2 glio_associate_resident");
    if (!c || slot < 0 || slot >= c->W) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int rc = glio_assoc_run(c, slot, q, t, out_count);
    if (rc == GLIO_OK) { c->have_factors = 1; c->f32_dirty = 1; }
    return rc;
}
// slide the window by one keyframe on the device: the resident scan of slot s+1 becomes that of slot s (slot W-1 is left for
// the next glio_set_scan); correspondences are not moved, the next glio_associate_window recomputes them (Q3: once per solve)
int glio_slide_window(glio_ctx* c) {
    if (!c) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    // the scans (and their presorted copies) stay where they are: the ring advances (round 4 moved 19 scans on the device and waited: 0.14 ms per keyframe)
    c->scan_base = (c->scan_base + 1) % c->W;
    for (int s = 0; s + 1 < c->W; ++s) c->h_scan_count[s] = c->h_scan_count[s + 1];
    c->h_scan_count[c->W - 1] = 0;
    CtxExtra* ex = extra_of(c);
    if (ex->ahead_valid) {          // glio_set_scan_ahead: the row that is slot W - 1 now already holds the new keyframe's scan (presorted) -- or will, behind this event
        GLIO_HIP_CHECK(hipStreamWaitEvent(c->stream, ex->ev_ahead, 0));
        c->h_scan_count[c->W - 1] = ex->ahead_n;
        ex->ahead_valid = 0;
    }
    return GLIO_OK;
}
int glio_select_correspondences(glio_ctx* c, int slot, const int32_t* indices, int n) {
    if (!c || slot < 0 || slot >= c->W || n < 0 || (n > 0 && !indices)) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    c->f32_dirty = 1;
    return glio_assoc_select(c, slot, indices, n);
}
int glio_select_correspondences_window(glio_ctx* c, const int32_t* offsets, const int32_t* indices, const uint8_t* changed) {
    if (!c || !offsets || (offsets[c->W] > 0 && !indices)) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    c->f32_dirty = 1;
    return glio_assoc_select_window(c, offsets, indices, changed);
}
int glio_associate_window(glio_ctx* c, const double* quats, const double* trans, int32_t* out_counts) {
    GLIO_TRACE("K2 glio_associate_window");
    if (!c || !quats || !trans) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int rc = glio_assoc_run_window(c, quats, trans, out_counts);
    if (rc == GLIO_OK) { c->have_factors = 1; c->f32_dirty = 1; }
    return rc;
}
int glio_associate_window_async(glio_ctx* c, const double* quats, const double* trans) {
    GLIO_TRACE("K2 glio_associate_window_async");
    if (!c || !quats || !trans) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int rc = glio_assoc_run_window_async(c, quats, trans);
    if (rc == GLIO_OK) { c->have_factors = 1; c->f32_dirty = 1; }
    return rc;
}
int glio_associate_window_counts(glio_ctx* c, int32_t* out_counts) {
    if (!c) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    // the caller has set this window's factor tables while the searches ran (the keyframe cycle's order): the solve's gather tables and the zeroed slices go
    // out now, behind the searches, instead of at the head of the solve with the GPU idle
    if (c->chain_tabs_dirty && c->have_factors) glio_chain_tabs_upload(c);
    const int rc = glio_assoc_finish_pending(c);
    if (rc != GLIO_OK) return rc;
    if (out_counts) for (int s = 0; s < c->W; ++s) out_counts[s] = c->h_count[s];
    return GLIO_OK;
}
int glio_associate(glio_ctx* c, int slot, const float* scan, int n, const double q[4], const double t[3], int* out_count) {
    GLIO_TRACE("K2 glio_associate");
    const int rc = glio_set_scan(c, slot, scan, n);
    if (rc != GLIO_OK) return rc;
    return glio_associate_resident(c, slot, q, t, out_count);
}

// ---------------------------------------------------------------------------------------------- IMU
static bool inv15(const double* A_in, double* Ainv) {
    double A[225];
    memcpy(A, A_in, sizeof A);
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) Ainv[i * 15 + j] = i == j;
    for (int c = 0; c < 15; ++c) {
        int piv = c; double best = fabs(A[c * 15 + c]);
        for (int r = c + 1; r < 15; ++r) if (fabs(A[r * 15 + c]) > best) { best = fabs(A[r * 15 + c]); piv = r; }
        if (best == 0.0) return false;
        if (piv != c) for (int j = 0; j < 15; ++j) { std::swap(A[c * 15 + j], A[piv * 15 + j]); std::swap(Ainv[c * 15 + j], Ainv[piv * 15 + j]); }
        const double d = A[c * 15 + c];
        for (int j = 0; j < 15; ++j) { A[c * 15 + j] /= d; Ainv[c * 15 + j] /= d; }
        for (int r = 0; r < 15; ++r) {
            if (r == c) continue;
            const double f = A[r * 15 + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 15; ++j) { A[r * 15 + j] -= f * A[c * 15 + j]; Ainv[r * 15 + j] -= f * Ainv[c * 15 + j]; }
        }
    }
    return true;
}
// sqrt_info = LLT(cov^-1).matrixL().transpose()  (ImuFactor.h:44-45) -- the reference recomputes this on
// the CPU in every Evaluate; it only depends on the pre-integration, so it is digested once at upload (Q5).
static bool digest_edge(const glio_preint* p, int slot, ImuEdgeDev* e);
extern "C++" bool glio_digest_imu_edge(const glio_preint* p, int slot, ImuEdgeDev* e) { return digest_edge(p, slot, e); }
static bool digest_edge(const glio_preint* p, int slot, ImuEdgeDev* e) {
    memset(e, 0, sizeof *e);
    memcpy(e->delta_p, p->delta_p, 24); memcpy(e->delta_q, p->delta_q, 32); memcpy(e->delta_v, p->delta_v, 24);
    memcpy(e->lin_ba, p->linearized_ba, 24); memcpy(e->lin_bg, p->linearized_bg, 24);
    e->sum_dt = p->sum_dt; e->slot_i = slot;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            e->dp_dba[r * 3 + c] = p->jacobian[(0 + r) * 15 + 9 + c];
            e->dp_dbg[r * 3 + c] = p->jacobian[(0 + r) * 15 + 12 + c];
            e->dq_dbg[r * 3 + c] = p->jacobian[(3 + r) * 15 + 12 + c];
            e->dv_dba[r * 3 + c] = p->jacobian[(6 + r) * 15 + 9 + c];
            e->dv_dbg[r * 3 + c] = p->jacobian[(6 + r) * 15 + 12 + c];
        }
    double I[225];
    if (!inv15(p->covariance, I)) return false;
    for (int j = 0; j < 15; ++j) {           // lower Cholesky of the (lower triangle of the) inverse
        double d = I[j * 15 + j];
        for (int k = 0; k < j; ++k) d -= I[j * 15 + k] * I[j * 15 + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        I[j * 15 + j] = d;
        for (int i = j + 1; i < 15; ++i) {
            double s = I[i * 15 + j];
            for (int k = 0; k < j; ++k) s -= I[i * 15 + k] * I[j * 15 + k];
            I[i * 15 + j] = s / d;
        }
    }
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) e->sqrt_info[i * 15 + j] = j >= i ? I[j * 15 + i] : 0.0;
    return true;
}

// ---------------------------------------------------------------------------------------------- pinned upload arena
// A set_* call stages every table in ONE pinned block [segment descriptors (1 KB) | payloads, 64-byte aligned]; stage_flush sends the
// block with ONE asynchronous copy to its device mirror and ONE kernel moves every segment to its destination (device-to-device
// segments -- results a kernel left in a workspace -- ride along): two operations per call instead of one copy per table (a keyframe
// cycle issued 25 copies of a few hundred bytes, ~5 us of GPU time and ~4 us of host time each).  The caller synchronises once.
#define STAGE_HEADER 1024
#define STAGE_DIRECT_BYTES 16384
struct StageSegDev { void* dst; const void* src; size_t bytes; };
__global__ __launch_bounds__(256) void k_unstage(const StageSegDev* __restrict__ segs) {
    const StageSegDev sg = segs[blockIdx.x];
    const size_t words = sg.bytes >> 2;          // every table is made of 4- or 8-byte items
    const unsigned int* __restrict__ src = static_cast<const unsigned int*>(sg.src);
    unsigned int* __restrict__ dst = static_cast<unsigned int*>(sg.dst);
    const size_t t = (size_t)blockIdx.y * 256 + threadIdx.x, nt = (size_t)gridDim.y * 256;
    size_t done = 0;
    if ((((size_t)sg.src | (size_t)sg.dst) & 15) == 0) {          // 16 bytes per thread and pass where both ends allow it, the tail by words
        const size_t quads = words >> 2;
        const uint4* __restrict__ s4 = static_cast<const uint4*>(sg.src);
        uint4* __restrict__ d4 = static_cast<uint4*>(sg.dst);
        for (size_t i = t; i < quads; i += nt) d4[i] = s4[i];
        done = quads << 2;
    }
    for (size_t i = done + t; i < words; i += nt) dst[i] = src[i];
}
#define UNSTAGE_GY 32
static int stage_reserve(glio_ctx* c, size_t bytes) {
    c->h_stage_used = STAGE_HEADER; c->n_stage_seg = 0;
    bytes += STAGE_HEADER + 64 * 34;
    c->h_stage_top = c->h_stage_cap & ~(size_t)63;
    if (bytes <= c->h_stage_cap) return GLIO_OK;
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_stage) hipFree(c->d_stage);
    c->h_stage = nullptr; c->d_stage = nullptr; c->h_stage_cap = 0;
    const size_t cap = bytes * 2 + 4096;
    GLIO_HIP_CHECK(hipHostMalloc((void**)&c->h_stage, cap));
    GLIO_HIP_CHECK(hipMalloc((void**)&c->d_stage, cap));
    c->h_stage_cap = cap; c->h_stage_top = cap & ~(size_t)63;
    return GLIO_OK;
}
// EARLY uploads.  glio_set_imu / glio_set_gnss are called while the window's searches are still running on the context's stream
// (findCorrespondingSurfFeaturesWindowAsync, then the factor tables): a copy enqueued on that stream would sit behind the searches, and the
// synchronisation that releases the pinned block would wait for all of them (0.1 ms of every keyframe cycle, the GPU idle behind the
// searches while the host then staged the GNSS tables).  So the block travels on a stream of its own into a device MIRROR that no kernel but
// k_unstage reads, the host waits for that copy alone, and k_unstage -- the only writer of the tables -- stays on the context's stream, behind the
// searches and every earlier reader of the tables, and waits for the copy through an event.  Two blocks take turns; a block is sent again only
// after the k_unstage that read its mirror (event ev_free, waited for by the upload stream).  GLIO_EARLY_UPLOAD=0: the in-stream path (A/B).
static bool stage_early_enabled(CtxExtra* ex) {
    if (ex->up_mode < 0) { const char* e = getenv("GLIO_EARLY_UPLOAD"); ex->up_mode = (!e || atoi(e) != 0) ? 1 : 0; }
    return ex->up_mode == 1;
}
static int stage_begin_early(glio_ctx* c, size_t bytes) {
    CtxExtra* ex = extra_of(c);
    if (!ex->up_stream) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) GLIO_HIP_CHECK(hipStreamCreateWithPriority(&ex->up_stream, hipStreamNonBlocking, greatest));
        else GLIO_HIP_CHECK(hipStreamCreateWithFlags(&ex->up_stream, hipStreamNonBlocking));
        GLIO_HIP_CHECK(hipEventCreateWithFlags(&ex->ev_up, hipEventDisableTiming));
    }
    const int k = ex->up_next;
    CtxExtra::UpArena& a = ex->up[k];
    bytes += STAGE_HEADER + 64 * 34;
    // the pinned block is written again: its last copy (two uploads ago) has to be through -- it has been for a long time; nothing else is waited for
    if (a.copying) { GLIO_HIP_CHECK(hipEventSynchronize(a.ev_copied)); a.copying = false; }
    if (!a.ev_copied) GLIO_HIP_CHECK(hipEventCreateWithFlags(&a.ev_copied, hipEventDisableTiming));
    if (bytes > a.cap) {
        if (a.pending) { GLIO_HIP_CHECK(hipEventSynchronize(a.ev_free)); a.pending = false; }
        if (a.h) hipHostFree(a.h);
        if (a.d) hipFree(a.d);
        a.h = nullptr; a.d = nullptr; a.cap = 0;
        const size_t cap = bytes * 2 + 4096;
        GLIO_HIP_CHECK(hipHostMalloc((void**)&a.h, cap));
        GLIO_HIP_CHECK(hipMalloc((void**)&a.d, cap));
        a.cap = cap;
        if (!a.ev_free) GLIO_HIP_CHECK(hipEventCreateWithFlags(&a.ev_free, hipEventDisableTiming));
    }
    // the staging functions work on the context's arena fields: point them at this block until stage_flush
    ex->sv_h = c->h_stage; ex->sv_d = c->d_stage; ex->sv_cap = c->h_stage_cap;
    c->h_stage = a.h; c->d_stage = a.d; c->h_stage_cap = a.cap;
    c->h_stage_used = STAGE_HEADER; c->n_stage_seg = 0; c->h_stage_top = a.cap & ~(size_t)63;
    ex->up_cur = k; ex->up_next = k ^ 1;
    return GLIO_OK;
}
static void stage_end_early(glio_ctx* c) {
    CtxExtra* ex = extra_of(c);
    if (ex->up_cur < 0) return;
    c->h_stage = ex->sv_h; c->d_stage = ex->sv_d; c->h_stage_cap = ex->sv_cap;
    c->h_stage_used = STAGE_HEADER; c->n_stage_seg = 0; c->h_stage_top = c->h_stage_cap & ~(size_t)63;
    ex->up_cur = -1;
}
static int stage_upload(glio_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return GLIO_OK;
    if (bytes > STAGE_DIRECT_BYTES && extra_of(c)->up_cur < 0) {          // a large table goes straight to its destination (the copy engine beats a second pass over it);
        const size_t need = (bytes + 63) & ~(size_t)63;      // its pinned copy is taken from the TOP of the arena, outside the block stage_flush sends
        if (c->h_stage_top < need || c->h_stage_top - need < c->h_stage_used) { glio_set_error("upload arena overflow"); return GLIO_E_STATE; }
        c->h_stage_top -= need;
        memcpy(c->h_stage + c->h_stage_top, src, bytes);
        GLIO_HIP_CHECK(hipMemcpyAsync(dst, c->h_stage + c->h_stage_top, bytes, hipMemcpyHostToDevice, c->stream));
        return GLIO_OK;
    }
    const size_t off = (c->h_stage_used + 63) & ~(size_t)63;
    if (off + bytes > c->h_stage_top || c->n_stage_seg >= 32 || (bytes & 3)) { glio_set_error("upload arena overflow"); return GLIO_E_STATE; }
    memcpy(c->h_stage + off, src, bytes);
    c->h_stage_used = off + bytes;
    c->stage_seg[c->n_stage_seg].dst = dst; c->stage_seg[c->n_stage_seg].src = c->d_stage + off; c->stage_seg[c->n_stage_seg].bytes = bytes;
    c->n_stage_seg += 1;
    return GLIO_OK;
}
// a device-to-device segment of the same batch
static int stage_d2d(glio_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return GLIO_OK;
    if (c->n_stage_seg >= 32 || (bytes & 3)) { glio_set_error("upload arena overflow"); return GLIO_E_STATE; }
    c->stage_seg[c->n_stage_seg].dst = dst; c->stage_seg[c->n_stage_seg].src = src; c->stage_seg[c->n_stage_seg].bytes = bytes;
    c->n_stage_seg += 1;
    return GLIO_OK;
}
static int stage_flush(glio_ctx* c) {
    const int n = c->n_stage_seg;
    CtxExtra* ex = extra_of(c);
    if (n == 0) { stage_end_early(c); return GLIO_OK; }
    static_assert(sizeof(StageSegDev) * 32 <= STAGE_HEADER, "descriptor header");
    memcpy(c->h_stage, c->stage_seg, sizeof(StageSegDev) * (size_t)n);
    if (ex->up_cur >= 0) {
        CtxExtra::UpArena& a = ex->up[ex->up_cur];
        hipError_t e = hipSuccess;
        if (a.pending) e = hipStreamWaitEvent(ex->up_stream, a.ev_free, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(a.d, a.h, c->h_stage_used, hipMemcpyHostToDevice, ex->up_stream);
        // k_unstage -- the only writer of the factor tables -- ON THE UPLOAD STREAM too, right behind its copy: every reader of those tables is an entry point
        // that has returned by the time glio_set_imu / glio_set_gnss can be called (solve, marginalization, linearisation and the evaluation calls all end with
        // a wait; what may still be in flight on the context's stream -- the window's searches, the local map, a presort -- reads none of them), and the next
        // reader on the context's stream waits for the event below.  Behind the searches on the context's stream instead (GLIO_UNSTAGE_IN_STREAM=1) the two
        // installs sat between the searches and the solve: 2 x (4 us + a 10 us cross-stream hand-over) of every keyframe call.
        if (ex->unstage_up < 0) { const char* w = getenv("GLIO_UNSTAGE_IN_STREAM"); ex->unstage_up = (w && atoi(w) != 0) ? 0 : 1; }
        if (ex->unstage_up) {
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_unstage, dim3(n, UNSTAGE_GY), dim3(256), 0, ex->up_stream, reinterpret_cast<const StageSegDev*>(a.d));
                e = hipEventRecord(a.ev_free, ex->up_stream);
                a.pending = true;
            }
            if (e == hipSuccess) e = hipEventRecord(ex->ev_up, ex->up_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ex->ev_up, 0);
        } else {
        if (e == hipSuccess) e = hipEventRecord(ex->ev_up, ex->up_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ex->ev_up, 0);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_unstage, dim3(n, UNSTAGE_GY), dim3(256), 0, c->stream, reinterpret_cast<const StageSegDev*>(a.d));
            e = hipEventRecord(a.ev_free, c->stream);
            a.pending = true;
        }
        }
        // the pinned block is this arena's until its next turn (stage_begin_early waits for the copy then): the call does not wait for the copy
        // (it did, with hipStreamSynchronize on the upload stream: 10-40 us of host time per call while a search kernel fills the chip; GLIO_EARLY_UPLOAD_WAIT=1)
        if (ex->up_wait < 0) { const char* w = getenv("GLIO_EARLY_UPLOAD_WAIT"); ex->up_wait = (w && atoi(w) != 0) ? 1 : 0; }
        if (e == hipSuccess) {
            if (ex->up_wait) e = hipStreamSynchronize(ex->up_stream);
            else { e = hipEventRecord(a.ev_copied, ex->up_stream); a.copying = true; }
        }
        stage_end_early(c);
        if (e != hipSuccess) { glio_set_error("early upload: %s", hipGetErrorString(e)); return GLIO_E_HIP; }
        return GLIO_OK;
    }
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_stage, c->h_stage, c->h_stage_used, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_unstage, dim3(n, UNSTAGE_GY), dim3(256), 0, c->stream, reinterpret_cast<const StageSegDev*>(c->d_stage));
    c->n_stage_seg = 0;
    return GLIO_OK;
}
#define STAGE(dst, src, bytes) do { const int rc_ = stage_upload(c, (dst), (src), (bytes)); if (rc_ != GLIO_OK) { stage_end_early(c); return rc_; } } while (0)

int glio_set_imu(glio_ctx* c, int n_edges, const glio_preint* edges, const int32_t* slot_i) {
    if (!c || n_edges < 0 || n_edges > c->W - 1 + (c->W == 1)) { glio_set_error("bad IMU edge count"); return GLIO_E_ARG; }
    { const int rp = marg_pending_done(c); if (rp) return rp; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    std::vector<ImuEdgeDev> h(std::max(1, n_edges));
    for (int k = 0; k < n_edges; ++k) {
        if (slot_i[k] < 0 || slot_i[k] + 1 >= c->W) { glio_set_error("IMU edge slot out of range"); return GLIO_E_ARG; }
        CtxExtra* ex = extra_of(c);
        int hit = -1;
        for (int j = k; j <= k + 1 && hit < 0; ++j)
            if (j < (int)ex->imu_raw.size() && memcmp(&ex->imu_raw[j], &edges[k], sizeof(glio_preint)) == 0) hit = j;
        if (hit >= 0) { h[k] = ex->imu_dig[hit]; h[k].slot_i = slot_i[k]; }
        else if (!digest_edge(&edges[k], slot_i[k], &h[k])) { glio_set_error("IMU covariance not invertible / not SPD"); return GLIO_E_NUMERIC; }
    }
    { CtxExtra* ex = extra_of(c); ex->imu_raw.assign(edges, edges + n_edges); ex->imu_dig.assign(h.begin(), h.begin() + n_edges); }
    if (n_edges) {
        const bool early = stage_early_enabled(extra_of(c));
        { const int rc = early ? stage_begin_early(c, n_edges * sizeof(ImuEdgeDev) + 64) : stage_reserve(c, n_edges * sizeof(ImuEdgeDev) + 64); if (rc != GLIO_OK) return rc; }
        STAGE(c->d_imu, h.data(), n_edges * sizeof(ImuEdgeDev));
        { const int rf = stage_flush(c); if (rf != GLIO_OK) return rf; }
        if (!early) GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    c->n_imu = n_edges;
    for (int k = 0; k < n_edges; ++k) c->h_imu_slot[k] = slot_i[k];
    c->chain_tabs_dirty = 1; c->h_band_clean = 0;
    extra_of(c)->imu_edge0 = -1;
    for (int k = 0; k < n_edges; ++k) if (slot_i[k] == 0) extra_of(c)->imu_edge0 = k;
    if (n_edges) c->have_factors = 1;
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- prior
int glio_set_prior(glio_ctx* c, const glio_prior* p) {
    if (!c) return GLIO_E_ARG;
    { const int rp = marg_pending_done(c); if (rp) return rp; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    if (!p || p->n <= 0) {
        c->prior_n = 0; c->prior_nb = 0; c->arrow.prior_ok = 1; c->arrow.prior_chain = 1;
        for (int k = 0; k < 15 * c->W; ++k) c->h_prior_index[k] = -1;
        c->chain_tabs_dirty = 1; c->h_band_clean = 0;
        return GLIO_OK;
    }
    const int np = p->n, nb = p->n_blocks, W = c->W;
    if (np > 6 * W + 9 || nb > 2 * W + 1) { glio_set_error("prior too large for window"); return GLIO_E_ARG; }
    std::vector<int> index(15 * W, -1), colblk(np, -1);
    for (int b = 0; b < nb; ++b) {
        const int s = p->blk_slot[b], k = p->blk_kind[b], idx = p->blk_idx[b];
        if (k != GLIO_BLK_TRANS && k != GLIO_BLK_QUAT && k != GLIO_BLK_SPEEDBIAS) { glio_set_error("prior block %d: unknown kind %d", b, k); return GLIO_E_ARG; }
        const int ls = k == GLIO_BLK_SPEEDBIAS ? 9 : 3;
        if (s < 0 || s >= W || idx < 0 || idx + ls > np) { glio_set_error("prior block out of range"); return GLIO_E_ARG; }
        const int off = 15 * s + (k == GLIO_BLK_TRANS ? 0 : (k == GLIO_BLK_QUAT ? 3 : 6));
        for (int j = 0; j < ls; ++j) {
            if (index[off + j] >= 0) { glio_set_error("prior: two blocks for slot %d kind %d", s, k); return GLIO_E_ARG; }
            if (colblk[idx + j] >= 0) { glio_set_error("prior: blocks %d and %d overlap at column %d", colblk[idx + j], b, idx + j); return GLIO_E_ARG; }
            index[off + j] = idx + j; colblk[idx + j] = b;
        }
    }
    for (int j = 0; j < np; ++j) if (colblk[j] < 0) { glio_set_error("prior column %d not covered by a block", j); return GLIO_E_ARG; }
    { int nsb = 0; for (int b = 0; b < nb; ++b) nsb += p->blk_kind[b] == GLIO_BLK_SPEEDBIAS; c->arrow.prior_ok = nsb <= 1; }   // two speed-bias blocks would couple the chain densely
    {   // Is J0^T J0 block diagonal by keyframe?  The reference's own marginalization always produces that (its LiDAR factors
        // constrain every pose absolutely, quirk Q7, so eliminating the oldest keyframe never couples two kept ones); then
        // the whole window is a keyframe chain and the solver needs no dense pose block.
        std::vector<double> cn(np, 0.0);
        for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) cn[j] += p->lin_jac[(size_t)i * np + j] * p->lin_jac[(size_t)i * np + j];
        bool chain = true;
        for (int a = 0; a < np && chain; ++a)
            for (int b2 = a + 1; b2 < np; ++b2) {
                if (p->blk_slot[colblk[a]] == p->blk_slot[colblk[b2]]) continue;
                double sab = 0;
                for (int i = 0; i < np; ++i) sab += p->lin_jac[(size_t)i * np + a] * p->lin_jac[(size_t)i * np + b2];
                if (fabs(sab) > 1e-13 * sqrt(cn[a] * cn[b2])) { chain = false; break; }
            }
        c->arrow.prior_chain = chain ? 1 : 0;
        c->prior_device_made = 0;
    }
    GnssDevExtra* ex = glio_extra(c);
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_J0, p->lin_jac, (size_t)np * np * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_r0, p->lin_res, np * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_x0, p->blk_x0, (size_t)nb * 9 * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_slot, p->blk_slot, nb * 4, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_kind, p->blk_kind, nb * 4, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_idx, p->blk_idx, nb * 4, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_prior_index, index.data(), 15 * W * 4, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(ex->d_prior_colblk, colblk.data(), np * 4, hipMemcpyHostToDevice));
    c->prior_n = np; c->prior_nb = nb;
    for (int k = 0; k < 15 * W; ++k) c->h_prior_index[k] = index[k];
    c->chain_tabs_dirty = 1; c->h_band_clean = 0;
    glio_launch_gram(c, np);
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_factors = 1;
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- GNSS
static void ecef2rotation_host(const double xyz[3], double R[9]) {   // gnss_utility.cpp:347-390,738-748
    const double e2 = 6.69437999014e-3, a = 6378137.0, R2D = 180.0 / M_PI, D2R = M_PI / 180.0;
    const double a2 = a * a, b2 = a2 * (1 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2;
    const double p = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
    double s1 = xyz[2] * a, s2 = p * b, h = sqrt(s1 * s1 + s2 * s2);
    const double st = s1 / h, ct = s2 / h;
    s1 = xyz[2] + ep2 * b * pow(st, 3);
    s2 = p - a * e2 * pow(ct, 3);
    const double lat = (atan(s1 / s2) * R2D) * D2R, lon = (atan2(xyz[1], xyz[0]) * R2D) * D2R;
    const double sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
    R[0] = -so; R[1] = -sl * co; R[2] = cl * co;
    R[3] = co;  R[4] = -sl * so; R[5] = cl * so;
    R[6] = 0;   R[7] = cl;       R[8] = sl;
}

int glio_set_gnss(glio_ctx* c, const glio_gnss_frame* frame, int n_dd, const glio_dd_psr* dd, int n_dop, const glio_doppler* dop) {
    if (!c || n_dd < 0 || n_dop < 0) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    { const int rp = marg_pending_done(c); if (rp) return rp; }
    const int W = c->W;
    if (frame) {
        c->frame = *frame;
        double Ree[9];
        ecef2rotation_host(frame->anc_ecef, Ree);
        const double s = sin(frame->yaw_enu_local), co = cos(frame->yaw_enu_local);
        const double Rel[9] = {co, -s, 0, s, co, 0, 0, 0, 1};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += Ree[i * 3 + k] * Rel[k * 3 + j];
            c->R_ecef_local[i * 3 + j] = a;
        }
    } else if (n_dd + n_dop > 0) { glio_set_error("GNSS factors need a frame"); return GLIO_E_ARG; }
    // sort by (slot_i, slot_j), Doppler additionally by epoch; group = one slot pair.  A caller that builds the factors keyframe pair by
    // keyframe pair (the reference's loop order, Estimator.cpp:2255-2421) hands them over sorted already: then nothing is copied or
    // moved on the host, the arrays are staged for upload as they are (0.17 -> 0.06 ms per keyframe at 152 + 1520 factors)
    struct Span { const glio_dd_psr* p; size_t n; const glio_dd_psr* begin() const { return p; } const glio_dd_psr* end() const { return p + n; }
                  size_t size() const { return n; } const glio_dd_psr& operator[](size_t k) const { return p[k]; } const glio_dd_psr* data() const { return p; } };
    struct SpanD { const glio_doppler* p; size_t n; const glio_doppler* begin() const { return p; } const glio_doppler* end() const { return p + n; }
                   size_t size() const { return n; } const glio_doppler& operator[](size_t k) const { return p[k]; } const glio_doppler* data() const { return p; } };
    std::vector<glio_dd_psr> dd_copy;
    std::vector<glio_doppler> dop_copy;
    Span sdd{dd, (size_t)n_dd};
    SpanD sdop{dop, (size_t)n_dop};
    for (auto& f : sdd) if (f.slot_i < 0 || f.slot_i >= W || f.slot_j < 0 || f.slot_j >= W || f.slot_i == f.slot_j || f.n_sat < 2 || f.n_sat > GLIO_DD_MAX_SAT || f.master < 0 || f.master >= f.n_sat) { glio_set_error("bad DD factor"); return GLIO_E_ARG; }
    for (auto& f : sdop) if (f.slot_i < 0 || f.slot_i >= W || f.slot_j < 0 || f.slot_j >= W || f.slot_i == f.slot_j || f.epoch < 0 || f.epoch >= c->n_ddt_max) { glio_set_error("bad Doppler factor (epoch %d, max_ddt_epochs %d)", f.epoch, c->n_ddt_max); return GLIO_E_ARG; }
    auto key = [W](int i, int j) { return i * W + j; };
    auto dd_less = [&](const glio_dd_psr& a, const glio_dd_psr& b) { return key(a.slot_i, a.slot_j) < key(b.slot_i, b.slot_j); };
    auto dop_less = [&](const glio_doppler& a, const glio_doppler& b) {
        const int ka = key(a.slot_i, a.slot_j), kb = key(b.slot_i, b.slot_j);
        return ka != kb ? ka < kb : a.epoch < b.epoch; };
    if (!std::is_sorted(sdd.begin(), sdd.end(), dd_less)) {
        dd_copy.assign(dd, dd + n_dd);
        std::stable_sort(dd_copy.begin(), dd_copy.end(), dd_less);
        sdd.p = dd_copy.data();
    }
    if (!std::is_sorted(sdop.begin(), sdop.end(), dop_less)) {
        dop_copy.assign(dop, dop + n_dop);
        std::stable_sort(dop_copy.begin(), dop_copy.end(), dop_less);
        sdop.p = dop_copy.data();
    }
    std::vector<GnssGroup> groups;
    std::vector<DopRun> runs;
    size_t a = 0, b = 0;
    while (a < sdd.size() || b < sdop.size()) {
        int k = 1 << 30;
        if (a < sdd.size()) k = std::min(k, key(sdd[a].slot_i, sdd[a].slot_j));
        if (b < sdop.size()) k = std::min(k, key(sdop[b].slot_i, sdop[b].slot_j));
        GnssGroup g;
        g.slot_i = k / W; g.slot_j = k % W;
        g.dd_begin = (int)a;
        while (a < sdd.size() && key(sdd[a].slot_i, sdd[a].slot_j) == k) ++a;
        g.dd_end = (int)a;
        g.dop_begin = (int)b;
        g.run_begin = (int)runs.size();
        while (b < sdop.size() && key(sdop[b].slot_i, sdop[b].slot_j) == k) {
            DopRun r;
            r.begin = (int)b; r.epoch = sdop[b].epoch; r.group = (int)groups.size();
            while (b < sdop.size() && key(sdop[b].slot_i, sdop[b].slot_j) == k && sdop[b].epoch == r.epoch) ++b;
            r.end = (int)b;
            runs.push_back(r);
        }
        g.dop_end = (int)b;
        g.run_end = (int)runs.size();
        groups.push_back(g);
    }
    // an epoch must belong to exactly one group (one clock-drift unknown per epoch, one bracketing pair)
    { std::vector<int> seen(std::max(1, c->n_ddt_max), 0); for (auto& r : runs) if (seen[r.epoch]++) { glio_set_error("epoch %d appears under two keyframe pairs", r.epoch); return GLIO_E_ARG; } }
    if ((int)groups.size() > W * W) return GLIO_E_ARG;
    if (sdd.size() > c->dd_cap || !c->d_dd) {
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_dd) { hipFree(c->d_dd); c->d_dd = nullptr; }
        c->dd_cap = sdd.size() + sdd.size() / 2 + 16;
        ALLOC(c->d_dd, c->dd_cap * sizeof(glio_dd_psr));
    }
    if (sdop.size() > c->dop_cap || !c->d_dop) {
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_dop) { hipFree(c->d_dop); c->d_dop = nullptr; }
        c->dop_cap = sdop.size() + sdop.size() / 2 + 16;
        ALLOC(c->d_dop, c->dop_cap * sizeof(glio_doppler));
    }
    // structure tables of the arrow solver: which keyframe slots each clock-drift epoch couples
    const int ne = std::max(1, c->n_ddt_max);
    std::vector<int2> es(ne, make_int2(-1, -1));
    std::vector<std::vector<int>> per(W);
    c->arrow.gnss_ok = 1; c->arrow.max_epoch = -1; c->arrow.gnss_chain = 1;
    for (auto& g : groups) if (g.slot_j - g.slot_i != 1) c->arrow.gnss_chain = 0;       // a pair that skips a keyframe (or is listed upper keyframe first) breaks the chain
    for (size_t k = 0; k < groups.size(); ++k) c->h_groups[k] = groups[k];
    c->chain_tabs_dirty = 1; c->h_band_clean = 0;
    for (auto& r : runs) {
        c->arrow.max_epoch = std::max(c->arrow.max_epoch, r.epoch);
        const GnssGroup& g = groups[r.group];
        const int lo = std::min(g.slot_i, g.slot_j), hi = std::max(g.slot_i, g.slot_j);
        es[r.epoch] = make_int2(lo, hi);
        per[lo].push_back(r.epoch); per[hi].push_back(r.epoch);
        if (hi - lo != 1) c->arrow.gnss_ok = 0;       // velocity coupling between non-adjacent keyframes: chain is not tridiagonal
    }
    std::vector<int> off(W + 1, 0), list;
    for (int i = 0; i < W; ++i) { off[i + 1] = off[i] + (int)per[i].size(); list.insert(list.end(), per[i].begin(), per[i].end()); }
    GnssDevExtra* ex = glio_extra(c);
    const bool early = stage_early_enabled(extra_of(c));
    {
        const size_t total = sdd.size() * sizeof(glio_dd_psr) + sdop.size() * sizeof(glio_doppler) + groups.size() * sizeof(GnssGroup) +
                             (size_t)ne * sizeof(int2) + (W + 1) * 4 + list.size() * 4 + runs.size() * sizeof(DopRun) + 8 * 64;
        const int rc = early ? stage_begin_early(c, total) : stage_reserve(c, total);
        if (rc != GLIO_OK) return rc;
    }
    STAGE(c->d_dd, sdd.data(), sdd.size() * sizeof(glio_dd_psr));
    STAGE(c->d_dop, sdop.data(), sdop.size() * sizeof(glio_doppler));
    STAGE(c->d_groups, groups.data(), groups.size() * sizeof(GnssGroup));
    STAGE(c->arrow.d_ep_slots, es.data(), (size_t)ne * sizeof(int2));
    STAGE(c->arrow.d_ep_off, off.data(), (size_t)(W + 1) * 4);
    STAGE(c->arrow.d_ep_list, list.data(), list.size() * 4);
    STAGE(ex->d_runs, runs.data(), runs.size() * sizeof(DopRun));
    ex->n_runs = (int)runs.size();
    { const int rf = stage_flush(c); if (rf != GLIO_OK) return rf; }
    GLIO_HIP_CHECK(hipMemsetAsync(c->d_ddt_blocks, 0, 2 * (size_t)std::max(1, c->n_ddt_max) * sizeof(DdtBlock), c->stream));
    if (!early) GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->n_dd = (int)sdd.size(); c->n_dop = (int)sdop.size(); c->n_groups = (int)groups.size();
    if (c->n_groups) c->have_factors = 1;
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- solve
static int check_state(glio_ctx* c, const glio_state* s) {
    if (!c || !s || !s->trans || !s->quat || !s->speed_bias) { glio_set_error("null state"); return GLIO_E_ARG; }
    if (s->n_ddt < 0 || s->n_ddt > c->n_ddt_max || (s->n_ddt > 0 && !s->rcv_ddt)) { glio_set_error("n_ddt %d exceeds max_ddt_epochs %d", s->n_ddt, c->n_ddt_max); return GLIO_E_ARG; }
    if (!c->have_factors) { glio_set_error("no factors set"); return GLIO_E_STATE; }
    // every Doppler epoch is an unknown of the problem: a state that carries fewer clock-drift slots than the factors
    // reference would leave x[16 W + epoch] unset and drop that epoch's column from H (silently wrong)
    if (c->n_dop > 0 && c->arrow.max_epoch >= s->n_ddt) {
        glio_set_error("Doppler factors reference clock-drift epoch %d but the state carries n_ddt = %d", c->arrow.max_epoch, s->n_ddt);
        return GLIO_E_ARG;
    }
    return GLIO_OK;
}
static void pack_state(glio_ctx* c, const glio_state* s, double* h) {
    const int W = c->W;
    memcpy(h, s->trans, 3 * W * 8); memcpy(h + 3 * W, s->quat, 4 * W * 8); memcpy(h + 7 * W, s->speed_bias, 9 * W * 8);
    if (s->n_ddt) memcpy(h + 16 * W, s->rcv_ddt, s->n_ddt * 8);
}
static void unpack_state(glio_ctx* c, const double* h, glio_state* s) {
    const int W = c->W;
    memcpy(s->trans, h, 3 * W * 8); memcpy(s->quat, h + 3 * W, 4 * W * 8); memcpy(s->speed_bias, h + 7 * W, 9 * W * 8);
    if (s->n_ddt) memcpy(s->rcv_ddt, h + 16 * W, s->n_ddt * 8);
}
// Development aid (GLIO_DEBUG_LDS_POISON=1): LDS is not cleared between kernels, so a kernel that reads LDS it has not
// written sees whatever ran on that CU before -- results that depend on the process history.  This fills the LDS of every
// CU with NaNs before each kernel group, which turns such a read into a loud, reproducible failure.
extern __shared__ double poison_lds[];
__global__ void k_lds_poison(int n) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) poison_lds[k] = __longlong_as_double(0x7ff8dead00000000ll + k);
}
static int g_poison = getenv("GLIO_DEBUG_LDS_POISON") ? atoi(getenv("GLIO_DEBUG_LDS_POISON")) : 0;
extern "C++" void glio_lds_poison_stream(hipStream_t stream) {       // also called by the batch solve (batch_tr_kernels.hip)
    if (!g_poison) return;
    static bool configured = false;
    if (!configured) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
    hipLaunchKernelGGL(k_lds_poison, dim3(1024), dim3(256), 160 * 1024, stream, 160 * 1024 / 8);
}
static void lds_poison(glio_ctx* c) { glio_lds_poison_stream(c->stream); }
// `dense`: also gather the blocks into the dense H, g (k_assemble).  The solve on the keyframe-chain path does not need it
// (k_chain_step reads the factor blocks); glio_linearize and the other solver paths do.
static void enqueue_linearize(glio_ctx* c, int use_status, int which, int n_ddt, int dense = 1) {
    if (c->chain_tabs_dirty) glio_chain_tabs_upload(c);      // gather tables + zeroed slices, ahead of the factor kernels that fill them
    c->want_pair_H = dense;
    lds_poison(c);
    if (c->merged_linearize) {
        glio_launch_linearize_all(c, use_status, which, n_ddt);
        if (dense) glio_launch_assemble(c, use_status, which, n_ddt, dense == 2);
        return;
    }
    glio_launch_lidar_linearize(c, use_status, which);
    glio_launch_small_factors(c, use_status, which, n_ddt);
    if (dense) glio_launch_assemble(c, use_status, which, n_ddt, dense == 2);
}
// status record + initial state, pinned host memory -> device, by a kernel of the solve's own queue: hipMemcpyAsync does the same with a blit kernel
// (~4 us) and a barrier packet behind it (another ~4 us before the first linearisation starts); this is one ~2 us launch with nothing behind it
__global__ __launch_bounds__(1024) void k_stage_in(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, const int nwords) {
    for (int k = threadIdx.x; k < nwords; k += 1024) dst[k] = __builtin_nontemporal_load(src + k);      // (one round of reads over the link for the usual window)
}
// enqueue one complete solve from the packed state in h_xbuf; no host synchronisation inside
static int enqueue_solve(glio_ctx* c, int n_ddt) {
    const int nx = glio_x_size(c->W, n_ddt);
    SolverStatus st;
    memset(&st, 0, sizeof st);
    st.cur = 1;                       // "candidate" buffer 0 holds the initial point
    st.cand_pending = 1;
    st.radius = c->opts.initial_trust_region_radius; st.mu = 1e-8; st.n_ddt = n_ddt; st.decrease_factor = 2.0;
    c->solve_id = c->solve_id % 30000 + 1;          // kernels of the previous solve's look-ahead group may still be draining:
    st.solve_id = c->solve_id;                      // their progress words carry the old tag and are ignored
    *c->h_status = st;
    hipLaunchKernelGGL(k_stage_in, dim3(1), dim3(1024), 0, c->stream, reinterpret_cast<unsigned long long*>(c->d_status),
                       reinterpret_cast<const unsigned long long*>(c->d_h_status), 64 + nx);                              // status + state
    // The trust-region loop lives on the device (SolverStatus); the host only feeds it kernel groups
    // [linearise, step].  Instead of queueing all max_iterations+1 groups blindly -- after convergence the rest are
    // empty launches, ~2.5 us each -- it enqueues group k + 1 when k_tr_prepare of group k reports (through a progress word in
    // mapped host memory) that the solve goes on, i.e. while group k's step (~70 us) is still running, and stops as soon
    // as the finished solve's tag shows up (`enqueue_lead` > 1 queues that many groups further ahead, < 1 all of them).
    const int id = c->solve_id;
    c->h_progress[2] = 0;            // a stop raised in an earlier solve must not outlive it (the ids wrap: a stale word would end the solve that reuses the id)
    c->solve_t0 = std::chrono::steady_clock::now();
    auto started = [&]() { const int w = c->h_progress[0]; return (w >> 16) == id ? (w & 0xffff) : 0; };
    const int total = c->opts.max_iterations + 1;
    const int lead = c->enqueue_lead < 1 ? total : c->enqueue_lead;
    const auto t_start = c->solve_t0;
    int enq = 0, spins = 0;
    const int dense = glio_chain_kind(c, n_ddt) == 3 ? 2 : glio_solver_needs_dense_H(c, n_ddt);      // (2: the band k_chain_solve<true> reads is enough)
    // options.max_solver_time_in_seconds: when the budget is spent the host raises the stop word; the state machine of the next
    // group sees it and ends the solve.  One more group is fed regardless of the look-ahead rule so that such a state machine runs.
    const bool timed = c->opts.max_solver_time_s > 0.0;
    const auto budget = std::chrono::duration<double>(timed ? c->opts.max_solver_time_s : 0.0);
    bool stop_sent = false, extra_group = false;
    while (enq < total) {
        if (c->h_progress[1] == id) break;
        if (timed && !stop_sent && std::chrono::steady_clock::now() - t_start > budget) {
            c->h_progress[2] = id;
            stop_sent = true; extra_group = true;
        }
        if (enq - started() < lead || extra_group) {
            extra_group = false;
            enqueue_linearize(c, 1, 0, n_ddt, dense);
            lds_poison(c);
            glio_launch_tr_step(c, n_ddt);
            ++enq;
        } else if (((++spins) & 0xfff) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(20)) {
            glio_set_error("solver made no progress for 20 s (group %d of %d)", started(), total);
            return GLIO_E_HIP;
        }
    }
    GLIO_HIP_CHECK(hipGetLastError());
    return GLIO_OK;
}
static void fill_summary(glio_ctx* c, glio_summary* sum) {
    if (!sum) return;
    const SolverStatus& st = *c->h_status;
    memset(sum, 0, sizeof *sum);
    sum->iterations = st.iteration; sum->successful_steps = st.successful; sum->termination = st.termination;
    for (int k = 0; k < c->W; ++k) sum->n_lidar_residuals += c->h_count[k];
    if (st.termination == GLIO_TERM_FAILURE) glio_set_error("trust-region solver failure (mu %g, iteration %d, invalid %d)", st.mu, st.iteration, st.invalid);
    sum->initial_cost = st.initial_cost; sum->final_cost = st.cost; sum->final_radius = st.radius; sum->gradient_max_norm = st.grad_max_norm;
}

int glio_solve(glio_ctx* c, glio_state* s, glio_summary* sum) {
    GLIO_TRACE("K3-K7 glio_solve (linearise + trust region, device resident)");
    int rc = check_state(c, s);
    if (rc) return rc;
    rc = marg_pending_done(c);
    if (rc) return rc;
    rc = glio_assoc_finish_pending(c);
    if (rc) return rc;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int n_ddt = s->n_ddt, nx = glio_x_size(c->W, n_ddt);
    pack_state(c, s, c->h_xbuf);
    rc = enqueue_solve(c, n_ddt);
    if (rc) return rc;
    // The kernel that ends the solve publishes status + state in mapped host memory and then the solve's tag: no
    // device-to-host copy, no stream synchronisation (the look-ahead group of empty launches drains behind our back; the
    // next call on this stream queues behind it).  Fallback after 20 s without the tag: the classic copy + sync.
    {
        const auto t_wait = std::chrono::steady_clock::now();
        int spins = 0;
        bool seen = true;
        // max_solver_time_s is also watched here: with every group queued ahead the enqueue loop ends before the budget can elapse
        const bool timed = c->opts.max_solver_time_s > 0.0;
        const auto budget = std::chrono::duration<double>(timed ? c->opts.max_solver_time_s : 0.0);
        while (c->h_progress[1] != c->solve_id) {
            if (((++spins) & 0x3f) == 0) {
                const auto now = std::chrono::steady_clock::now();
                if (timed && c->h_progress[2] != c->solve_id && now - c->solve_t0 > budget) c->h_progress[2] = c->solve_id;
                if (now - t_wait > std::chrono::seconds(20)) { seen = false; break; }
                if (now - t_wait > std::chrono::milliseconds(2)) std::this_thread::yield();     // a long solve (C5-sized window): stop monopolising the core
            }
        }
        if (seen) {
            // the tag is there; the payload is accepted when its checksum adds up (glio_device.h: under load the tag has been seen
            // ahead of parts of the payload).  Re-read for up to 2 ms, then take the stream-ordered copy.
            const auto t_chk = std::chrono::steady_clock::now();
            for (;;) {
                // a PRIVATE copy, taken word by word with 8-byte atomic loads (the device writes naturally aligned 8-byte words; neither side can
                // tear one), and only the private copy is validated and used: the happens-before argument is in DESIGN.md section 5
                {
                    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(c->h_result);
                    unsigned long long* ds = reinterpret_cast<unsigned long long*>(c->h_status);
                    for (size_t w = 0; w < sizeof(SolverStatus) / 8; ++w) ds[w] = __atomic_load_n(src + w, __ATOMIC_RELAXED);
                    unsigned long long* dx = reinterpret_cast<unsigned long long*>(c->h_xbuf);
                    for (int k = 0; k < nx; ++k) dx[k] = __atomic_load_n(src + 64 + k, __ATOMIC_RELAXED);
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
                SolverStatus t = *c->h_status;
                const unsigned long long want = t.checksum;
                t.checksum = 0;
                unsigned long long sum = 0;
                const unsigned long long* words = reinterpret_cast<const unsigned long long*>(&t);
                for (size_t w = 0; w < sizeof(SolverStatus) / 8; ++w) sum += glio_result_mix(words[w], (unsigned long long)w);
                const unsigned long long* xw = reinterpret_cast<const unsigned long long*>(c->h_xbuf);
                for (int k = 0; k < nx; ++k) sum += glio_result_mix(xw[k], 64ull + (unsigned long long)k);
                if (sum == want && t.solve_id == c->solve_id && t.done) break;
                if (std::chrono::steady_clock::now() - t_chk > std::chrono::milliseconds(2)) { seen = false; break; }
            }
        }
        if (!seen) {
            GLIO_HIP_CHECK(hipMemcpyAsync(c->h_status, c->d_status, sizeof(SolverStatus), hipMemcpyDeviceToHost, c->stream));
            GLIO_HIP_CHECK(hipMemcpyAsync(c->h_xbuf, c->d_xout, (size_t)nx * 8, hipMemcpyDeviceToHost, c->stream));
            GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
    }
    fill_summary(c, sum);
    if (!c->h_status->done) { glio_set_error("solver did not finish"); return GLIO_E_STATE; }
    unpack_state(c, c->h_xbuf, s);
    c->last_n_ddt = n_ddt;
    if (c->h_status->termination == GLIO_TERM_FAILURE) return GLIO_E_NUMERIC;
    return GLIO_OK;
}

int glio_linearize(glio_ctx* c, const glio_state* s, double* H, double* g, double* cost) {
    GLIO_TRACE("K3-K6 glio_linearize");
    int rc = check_state(c, s);
    if (rc) return rc;
    rc = marg_pending_done(c);
    if (rc) return rc;
    rc = glio_assoc_finish_pending(c);
    if (rc) return rc;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int n_ddt = s->n_ddt, nx = glio_x_size(c->W, n_ddt), n = 15 * c->W + n_ddt;
    pack_state(c, s, c->h_xbuf);
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_x[0], c->h_xbuf, (size_t)nx * 8, hipMemcpyHostToDevice, c->stream));
    enqueue_linearize(c, 0, 0, n_ddt);
    GLIO_HIP_CHECK(hipGetLastError());
    if (H) GLIO_HIP_CHECK(hipMemcpyAsync(H, c->d_H[0], (size_t)n * n * 8, hipMemcpyDeviceToHost, c->stream));
    if (g) GLIO_HIP_CHECK(hipMemcpyAsync(g, c->d_g[0], (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (cost) GLIO_HIP_CHECK(hipMemcpyAsync(cost, c->d_cost[0], 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->last_n_ddt = n_ddt;
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- marginalization
int glio_marginalize(glio_ctx* c, const glio_state* s, double* lin_jac, double* lin_res, int32_t* blk_slot, int32_t* blk_kind,
                     int32_t* blk_idx, double* blk_x0, int32_t* out_n, int32_t* out_n_blocks) {
    GLIO_TRACE("glio_marginalize");
    { const int rp = marg_pending_done(c); if (rp) return rp; }
    int rc = check_state(c, s);
    if (rc) return rc;
    rc = glio_assoc_finish_pending(c);
    if (rc) return rc;
    if (!lin_jac || !lin_res || !blk_slot || !blk_kind || !blk_idx || !blk_x0) { glio_set_error("null output"); return GLIO_E_ARG; }
    const int W = c->W;
    if (W < 2) { glio_set_error("marginalization needs a window of at least 2 keyframes"); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int n_ddt = s->n_ddt, nx = glio_x_size(W, n_ddt), n = 6 * (W - 1) + 9;
    pack_state(c, s, c->h_xbuf);
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_x[0], c->h_xbuf, (size_t)nx * 8, hipMemcpyHostToDevice, c->stream));
    lds_poison(c);
    glio_launch_lidar_linearize(c, 0, 0, 1); glio_launch_lidar_reduce(c, 0);
    lds_poison(c);
    glio_launch_small_factors(c, 0, 0, n_ddt, 1);
    lds_poison(c);
    double *dJ, *dr; int* dok;
    glio_launch_marginalize(c, extra_of(c)->imu_edge0, &dJ, &dr, &dok);
    GLIO_HIP_CHECK(hipGetLastError());
    int ok = 0;
    GLIO_HIP_CHECK(hipMemcpyAsync(lin_jac, dJ, (size_t)n * n * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(lin_res, dr, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(&ok, dok, 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!ok) { glio_set_error("marginalization: Schur complement is not positive definite (rank-deficient kept block)"); return GLIO_E_NUMERIC; }
    // GetParameterBlocks with addr_shift slot s -> s-1 (Estimator.cpp:2584-2600)
    int nb = 0;
    for (int k = 1; k < W; ++k) {
        const int kinds = k == 1 ? 3 : 2;
        for (int kind = 0; kind < kinds; ++kind) {
            blk_slot[nb] = k - 1; blk_kind[nb] = kind;
            blk_idx[nb] = k == 1 ? (kind == 0 ? 0 : (kind == 1 ? 3 : 6)) : 15 + 6 * (k - 2) + 3 * kind;
            const double* src = kind == 0 ? s->trans + 3 * k : (kind == 1 ? s->quat + 4 * k : s->speed_bias + 9 * k);
            const int gs = kind == 0 ? 3 : (kind == 1 ? 4 : 9);
            for (int j = 0; j < 9; ++j) blk_x0[9 * nb + j] = j < gs ? src[j] : 0.0;
            ++nb;
        }
    }
    if (out_n) *out_n = n;
    if (out_n_blocks) *out_n_blocks = nb;
    return GLIO_OK;
}

// Marginalize and KEEP: the result becomes this context's prior for the next window without leaving the device (J0, r0 are
// copied device to device; only the small block tables are rebuilt on the host).  The caller slides its state arrays by one
// keyframe afterwards.  Equivalent to glio_marginalize + glio_set_prior(result), minus two PCIe trips of the n x n matrix.
static int marginalize_keep_wait(glio_ctx* c);
// an asynchronous marginalization nobody finished: every entry point that reads the prior, or writes the pinned arena its flag lives in, finishes it first
static int marg_pending_done(glio_ctx* c) { return (c && extra_of(c)->marg_h_ok) ? marginalize_keep_wait(c) : GLIO_OK; }
int glio_marginalize_keep(glio_ctx* c, const glio_state* s) {
    const int rc = glio_marginalize_keep_async(c, s);
    return rc != GLIO_OK ? rc : glio_marginalize_keep_finish(c);
}
int glio_marginalize_keep_finish(glio_ctx* c) {
    if (!c) return GLIO_E_ARG;
    if (!extra_of(c)->marg_h_ok) return GLIO_OK;            // nothing pending
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    return marginalize_keep_wait(c);
}
int glio_marginalize_keep_async(glio_ctx* c, const glio_state* s) {
    GLIO_TRACE("glio_marginalize_keep");
    int rc = check_state(c, s);
    if (rc) return rc;
    rc = marg_pending_done(c);
    if (rc) return rc;
    rc = glio_assoc_finish_pending(c);
    if (rc) return rc;
    const int W = c->W;
    if (W < 2) { glio_set_error("marginalization needs a window of at least 2 keyframes"); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int n_ddt = s->n_ddt, nx = glio_x_size(W, n_ddt), n = 6 * (W - 1) + 9, nb = 2 * (W - 1) + 1;
    pack_state(c, s, c->h_xbuf);
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_x[0], c->h_xbuf, (size_t)nx * 8, hipMemcpyHostToDevice, c->stream));
    lds_poison(c);
    glio_launch_lidar_linearize(c, 0, 0, 1); glio_launch_lidar_reduce(c, 0);
    lds_poison(c);
    glio_launch_small_factors(c, 0, 0, n_ddt, 1);
    lds_poison(c);
    double *dJ, *dr; int* dok;
    glio_launch_marginalize(c, extra_of(c)->imu_edge0, &dJ, &dr, &dok);
    GLIO_HIP_CHECK(hipGetLastError());
    // the new prior's block tables (slots already shifted s -> s-1, Estimator.cpp:2584-2600).  Everything below is enqueued behind the
    // marginalization kernels without waiting for them: the tables go through the pinned arena, the "positive definite" flag is
    // read back last, ONE synchronisation ends the call (two synchronisations and eight pageable copies cost ~0.1 ms of the 0.33 ms)
    std::vector<int> slot(nb), kind(nb), idx(nb), index(15 * W, -1), colblk(n, -1);
    std::vector<double> x0((size_t)nb * 9, 0.0);
    int b = 0;
    for (int k = 1; k < W; ++k) {
        const int kinds = k == 1 ? 3 : 2;
        for (int kd = 0; kd < kinds; ++kd, ++b) {
            slot[b] = k - 1; kind[b] = kd;
            idx[b] = k == 1 ? (kd == 0 ? 0 : (kd == 1 ? 3 : 6)) : 15 + 6 * (k - 2) + 3 * kd;
            const double* src = kd == 0 ? s->trans + 3 * k : (kd == 1 ? s->quat + 4 * k : s->speed_bias + 9 * k);
            const int gs = kd == 0 ? 3 : (kd == 1 ? 4 : 9), ls = kd == 2 ? 9 : 3;
            for (int j = 0; j < gs; ++j) x0[9 * b + j] = src[j];
            const int off = 15 * (k - 1) + (kd == 0 ? 0 : (kd == 1 ? 3 : 6));
            for (int j = 0; j < ls; ++j) { index[off + j] = idx[b] + j; colblk[idx[b] + j] = b; }
        }
    }
    GnssDevExtra* ex = glio_extra(c);
    { const int rs = stage_reserve(c, (size_t)nb * 9 * 8 + 3 * (size_t)nb * 4 + 15 * (size_t)W * 4 + (size_t)n * 4 + 8 * 64 + 64); if (rs != GLIO_OK) return rs; }
    { int rd = stage_d2d(c, c->d_prior_J0, dJ, (size_t)n * n * 8); if (rd == GLIO_OK) rd = stage_d2d(c, c->d_prior_r0, dr, (size_t)n * 8); if (rd != GLIO_OK) return rd; }
    STAGE(c->d_prior_x0, x0.data(), (size_t)nb * 9 * 8);
    STAGE(c->d_prior_slot, slot.data(), (size_t)nb * 4);
    STAGE(c->d_prior_kind, kind.data(), (size_t)nb * 4);
    STAGE(c->d_prior_idx, idx.data(), (size_t)nb * 4);
    STAGE(c->d_prior_index, index.data(), (size_t)15 * W * 4);
    STAGE(ex->d_prior_colblk, colblk.data(), (size_t)n * 4);
    { const int rf = stage_flush(c); if (rf != GLIO_OK) return rf; }
    // a Schur complement of block-diagonal pieces (LiDAR blocks, the IMU edge of the dropped keyframe, a block-diagonal old
    // prior) is block diagonal, and so is its Cholesky root: the chain property is inherited
    const int chain = c->prior_n > 0 ? c->arrow.prior_chain : 1;
    c->prior_n = n; c->prior_nb = nb;
    for (int k = 0; k < 15 * W; ++k) c->h_prior_index[k] = index[k];
    c->chain_tabs_dirty = 1; c->h_band_clean = 0;
    c->arrow.prior_ok = 1; c->arrow.prior_chain = chain;
    c->prior_device_made = 1;
    glio_launch_gram(c, n);
    int* h_ok = reinterpret_cast<int*>(c->h_stage + ((c->h_stage_used + 63) & ~(size_t)63));      // (pinned; reserved above)
    GLIO_HIP_CHECK(hipMemcpyAsync(h_ok, dok, 4, hipMemcpyDeviceToHost, c->stream));
    extra_of(c)->marg_h_ok = h_ok;                         // glio_marginalize_keep_finish (or the entry points that use the prior) waits and looks at it
    return GLIO_OK;
}
static int marginalize_keep_wait(glio_ctx* c) {
    const int W = c->W;
    int* h_ok = extra_of(c)->marg_h_ok;
    extra_of(c)->marg_h_ok = nullptr;
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!*h_ok) {          // the installed tables describe a factor that does not exist: the context is left WITHOUT a prior
        c->prior_n = 0; c->prior_nb = 0; c->arrow.prior_ok = 1; c->arrow.prior_chain = 1;
        for (int k = 0; k < 15 * W; ++k) c->h_prior_index[k] = -1;
        c->chain_tabs_dirty = 1; c->h_band_clean = 0;
        glio_set_error("marginalization: Schur complement is not positive definite (the context now has no prior)");
        return GLIO_E_NUMERIC;
    }
    return GLIO_OK;
}

}  // extern "C"
// R_ecef_local = R_ecef_enu(anchor) * Rz(yaw)  (dd_psr_factor.hpp:33-45) -- shared with eval_kernels.hip
void glio_host_ecef_local(const double anc[3], double yaw, double R[9]) {
    double Ree[9];
    ecef2rotation_host(anc, Ree);
    const double s = sin(yaw), co = cos(yaw);
    const double Rel[9] = {co, -s, 0, s, co, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += Ree[i * 3 + k] * Rel[k * 3 + j];
        R[i * 3 + j] = a;
    }
}
extern "C" {
// ---------------------------------------------------------------------------------------------- evaluators
int glio_eval_lidar_plane(glio_ctx* c, const float cp[4], const float plane[4], double score,
                          double const* const* P, double* res, double** J) {
    if (!c || !P || !res) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    CtxExtra* ex = extra_of(c);
    double h[7];
    memcpy(h, P[0], 24); memcpy(h + 3, P[1], 32);
    GLIO_HIP_CHECK(hipMemcpyAsync(ex->d_eval_params, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    glio_launch_eval_lidar(c, cp, plane, score, ex->d_eval_params, ex->d_eval_out);
    GLIO_HIP_CHECK(hipMemcpyAsync(ex->h_eval, ex->d_eval_out, 8 * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    res[0] = ex->h_eval[0];
    if (J) {
        if (J[0]) memcpy(J[0], ex->h_eval + 1, 24);
        if (J[1]) memcpy(J[1], ex->h_eval + 4, 32);
    }
    return GLIO_OK;
}

int glio_eval_imu(glio_ctx* c, const glio_preint* pre, double const* const* P, double* res, double** J) {
    if (!c || !pre || !P || !res) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    CtxExtra* ex = extra_of(c);
    ImuEdgeDev e;
    if (!digest_edge(pre, 0, &e)) { glio_set_error("IMU covariance not invertible / not SPD"); return GLIO_E_NUMERIC; }
    double h[32];
    const int sz[6] = {3, 4, 9, 3, 4, 9};
    int off = 0;
    for (int b = 0; b < 6; ++b) { memcpy(h + off, P[b], sz[b] * 8); off += sz[b]; }
    GLIO_HIP_CHECK(hipMemcpyAsync(ex->d_eval_edge, &e, sizeof e, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(ex->d_eval_params, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));   // e and h live on the stack
    glio_launch_eval_imu(c, ex->d_eval_edge, ex->d_eval_params, ex->d_eval_out);
    GLIO_HIP_CHECK(hipMemcpyAsync(ex->h_eval, ex->d_eval_out, (15 + 15 * 32) * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(res, ex->h_eval, 15 * 8);
    if (J) {
        off = 0;
        for (int b = 0; b < 6; ++b) {
            if (J[b]) for (int r = 0; r < 15; ++r) for (int k = 0; k < sz[b]; ++k) J[b][r * sz[b] + k] = ex->h_eval[15 + r * 32 + off + k];
            off += sz[b];
        }
    }
    return GLIO_OK;
}

// solver selection for tests: 0 = dense Cholesky only, 1 = structured (arrow) factorisation when the graph permits
int glio_debug_set_solver(glio_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 4) return GLIO_E_ARG;      // 0 dense only, 1 structured when possible, 2 = 1 + the chain kernel reports a
    c->arrow.mode = mode;                                   // breakdown every time (test hook for its dense fallback), 3 = 1 with the
                                                            // legacy chain sequence (assemble + k_chain_solve + k_tr_finish) instead of k_chain_step,
                                                            // 4 = 1 without the chain kernels (the arrow factorisation although the graph is a chain)
    return GLIO_OK;
}
int glio_debug_arrow_stamps(glio_ctx* c, long long* out64) {
    if (!c || !out64) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    GLIO_HIP_CHECK(hipMemcpy(out64, c->arrow.d_dbg, 320 * 8, hipMemcpyDeviceToHost));
    return GLIO_OK;
}
int glio_debug_set_merged_linearize(glio_ctx* c, int on) { if (!c) return GLIO_E_ARG; c->merged_linearize = on; return GLIO_OK; }
// kernel groups kept queued ahead of the GPU by glio_solve (0 = queue all max_iterations+1 groups up front)
int glio_debug_set_enqueue_lead(glio_ctx* c, int lead) {
    if (!c || lead < 0) return GLIO_E_ARG;
    c->enqueue_lead = lead;
    return GLIO_OK;
}
int glio_debug_solver_path(glio_ctx* c) { return c ? c->arrow.last_path : -1; }
int glio_debug_chain_fronts_used(glio_ctx* c) { return c ? c->arrow.last_fronts : -1; }
int glio_debug_set_k3(glio_ctx* c, int bpk, int unroll) {
    if (!c || bpk < 1 || bpk > GLIO_K3_MAX_BLOCKS_PER_KF || (unroll != 1 && unroll != 2 && unroll != 4 && unroll != 8 && unroll != 12 && unroll != 14 && unroll != 18 && unroll != 21 && unroll != 22 && unroll != 24 && unroll != 32 && unroll != 33 && unroll != 34 && unroll != 35)) return GLIO_E_ARG;
    c->k3_bpk = bpk; c->k3_unroll = unroll;
    return GLIO_OK;
}

// ---------------------------------------------------------------------------------------------- timing hooks
int glio_time_kernel(glio_ctx* c, int which, int reps, float* ms_out) {
    if (!c || reps < 1 || !ms_out) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    if (which == GLIO_KERNEL_ASSOCIATE || which == GLIO_KERNEL_MAP_BUILD) { glio_assoc_time_hooks(c, which, reps, ms_out); return GLIO_OK; }
    if (!c->have_factors) return GLIO_E_STATE;
    const int n_ddt = c->last_n_ddt;
    // warm-up launch, then `reps` timed launches on the context's stream
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass == 0 ? 1 : reps;
        if (pass == 1) GLIO_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
        for (int k = 0; k < r; ++k) {
            if (which == GLIO_KERNEL_LIDAR_LINEARIZE) glio_launch_lidar_linearize(c, 0, 0);
            else if (which == GLIO_KERNEL_STREAM_READ) glio_launch_stream_read(c);
            else if (which == GLIO_KERNEL_FULL_LINEARIZE) enqueue_linearize(c, 0, 0, n_ddt);
            else if (which == GLIO_KERNEL_LINEARIZE_ALL) { c->want_pair_H = glio_solver_needs_dense_H(c, n_ddt); glio_launch_linearize_all(c, 0, 0, n_ddt); }   // as launched inside the solve
            else if (which == GLIO_KERNEL_TR_STEP) {
                // one first-iteration step computation (scale, Cauchy, Cholesky, dogleg) on H[0]
                SolverStatus st;
                memset(&st, 0, sizeof st);
                st.cur = 1; st.cand_pending = 1; st.radius = c->opts.initial_trust_region_radius; st.mu = 1e-8; st.n_ddt = n_ddt;
                *c->h_status = st;
                GLIO_HIP_CHECK(hipMemcpyAsync(c->d_status, c->h_status, sizeof st, hipMemcpyHostToDevice, c->stream));
                glio_launch_tr_step(c, n_ddt);
            } else if (which == GLIO_KERNEL_TR_STEP_STEADY) {
                // a later step: the candidate in buffer 0 is pending and will be ACCEPTED (a cost far above it, a positive model change), the Jacobi scale of the
                // last solve is in place, mu at its floor -- the step the helpers' speculative build applies to
                SolverStatus st;
                memset(&st, 0, sizeof st);
                st.cur = 1; st.cand_pending = 1; st.phase = 1; st.iteration = 1; st.radius = c->opts.initial_trust_region_radius; st.mu = 1e-8; st.n_ddt = n_ddt;
                st.cost = 1e30; st.initial_cost = 1e30; st.model_cost_change = 1.0; st.dogleg_step_norm = 1.0;
                *c->h_status = st;
                GLIO_HIP_CHECK(hipMemcpyAsync(c->d_status, c->h_status, sizeof st, hipMemcpyHostToDevice, c->stream));
                glio_launch_tr_step(c, n_ddt);
            } else if (which == GLIO_KERNEL_MARGINALIZE) {
                if (c->W < 2) return GLIO_E_ARG;
                double *dJ, *dr; int* dok;
                glio_launch_lidar_linearize(c, 0, 0, 1); glio_launch_lidar_reduce(c, 0);
                glio_launch_small_factors(c, 0, 0, n_ddt, 1);
                glio_launch_marginalize(c, extra_of(c)->imu_edge0, &dJ, &dr, &dok);
            } else return GLIO_E_ARG;
        }
        if (pass == 1) GLIO_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    float ms = 0;
    GLIO_HIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_out = ms / reps;
    return GLIO_OK;
}

int glio_time_solve(glio_ctx* c, const glio_state* s, int reps, float* ms_out, glio_summary* last) {
    int rc = check_state(c, s);
    if (rc) return rc;
    if (reps < 1 || !ms_out) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    const int n_ddt = s->n_ddt;
    pack_state(c, s, c->h_xbuf);
    GLIO_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < reps; ++k) {
        rc = enqueue_solve(c, n_ddt);
        if (rc) return rc;
        // h_status is re-read by the next enqueue's async H2D: wait for that copy, not for the solve
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    GLIO_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(c->h_status, c->d_status, sizeof(SolverStatus), hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    float ms = 0;
    GLIO_HIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_out = ms / reps;
    fill_summary(c, last);
    c->last_n_ddt = n_ddt;
    return GLIO_OK;
}

}  // extern "C"
