// sv_hip.hip — C ABI of libsv_hip (include/sv_hip.h): context, buffers, launch order.
//
// Launch order per frame = the stage order of compute_disparity
// (/root/reference/modules/DisparityModule/sgbm_cuda/bova_sgm.cu:87-136) without its
// cudaDeviceSynchronize() after every kernel, its two volume memsets and its blocking copies:
// everything is enqueued on one HIP stream; the host variant synchronises once at the end.
#include "../../include/sv_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "sv_kernels.hip.h"
#include "sv_sweep.hip.h"
#include "sv_blf.hip.h"

namespace {

struct Buf {
    void* p = nullptr;
    size_t n = 0;
};

}  // namespace

// Scratch of sv_filter_speckles_static_dev, the one entry point that allocates nothing: labels and
// component sizes of the union-find speckle filter, one int each per pixel, for frames of up to
// SV_STATIC_MAX_PIXELS pixels (16.7 MB; every GPU the library runs on holds its own copy until the
// process ends).  In host code this name is only a handle for hipGetSymbolAddress: the address the
// kernels get is g_scratch_dev[device], looked up once per device in sv_create with that device current.
static __device__ __align__(256) int sv_static_scratch[2 * SV_STATIC_MAX_PIXELS];
namespace {
constexpr int MAX_DEVICES = 64;
int* g_scratch_dev[MAX_DEVICES] = {};   // per device: device address of its copy of sv_static_scratch
std::mutex g_scratch_mu;

// called with `device` current
hipError_t static_scratch_lookup(int device, int** out) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_dev[device]) {
        int* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(sv_static_scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_dev[device] = ptr;
    }
    *out = g_scratch_dev[device];   // the context keeps its own copy: no entry point reads the table without the mutex
    return hipSuccess;
}
}  // namespace

struct sv_ctx {
    int device = 0;
    int* static_scratch = nullptr;     // this device's copy of sv_static_scratch (hipGetSymbolAddress in sv_create)
    int W = 0, H = 0;
    int batch = 1;  // frames per launch the volume buffers are sized for
    bool core_ok = false;  // the core buffers (statistics, C, S, records, planes) are allocated for `batch` frames
    sv_params prm{};
    hipStream_t stream = nullptr;
    char err[512] = {0};

    // device buffers
    Buf img_l, img_r;        // u8 [H][W] staging for the host entry points
    Buf stat_l, stat_r;      // left: uint2 [H][W] records; right: u16 planes [3][2][H][pitch], two 8-bit statistics per element
    Buf C, S;                // u16 [H][W1][D]
    Buf rec;                 // WTA records {disp16, raw, minS, 0} [H][W] x 8 bytes
    bool debug = false;      // materialise S and run WTA as its own kernel (sv_debug_read_volume_host)
    int debug_mode = 0;      // as given to sv_set_debug
    bool force_fuse = false; // sv_set_debug(ctx, 2): WTA fused into the last path whatever the batch size
    // asynchronous host-buffer pipeline (sv_submit_gray_host / sv_wait_host): pinned staging slots
    static constexpr int N_SLOTS = SV_MAX_IN_FLIGHT;
    struct Slot {
        uint8_t* in = nullptr;    // pinned: left image, right image (dense)
        int16_t* out = nullptr;   // pinned: disparity16 (dense)
        hipEvent_t done = nullptr;
        bool busy = false;
        int rc = 0;
    } slot[N_SLOTS];
    bool slots_ok = false;
    long next_ticket = 0;
    // batched asynchronous host pipeline (sv_submit_gray_batch_host / sv_wait_batch_host)
    struct BatchSlot {
        Buf in_l, in_r, out;      // device: [n][H][W] u8 x 2, [n][H][W] i16
        hipEvent_t up = nullptr, computed = nullptr, done = nullptr;
        bool busy = false;
        std::vector<int16_t*> dst;  // where the batch's frames go on the host (kept for the retry in sv_wait_batch_host)
        size_t ds = 0;
    } bslot[SV_BATCH_SLOTS];
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    bool bslots_ok = false;
    long next_batch_ticket = 0;
    // fused four-path sweeps (sv_sweep.hip.h)
    Buf sw_handoff, sw_progress, sw_sink, sw_prof, cost_sink;
    int sw_prof_n = 0;
    unsigned int* sw_error = nullptr;  // pinned host word the sweep kernels set when a bounded wait runs out
    int sweep_mode = 0;                // 0 auto (batch >= sweep_min_batch), 1 always (when applicable), -1 never
    int sweep_min_batch = 0;           // > 0: fixed threshold (SV_SWEEP_MIN_BATCH); 0: one frame per workgroup pipeline
    int occ_sweep[2][8][2] = {};       // workgroups of a sweep kernel instance a CU holds (0: not asked yet, -1: the query failed)
    bool attr_cost2[2][3][8] = {}, attr_lrcheck = false;
    int sweep_cw = 0;                  // compute waves per sweep workgroup: 15 (one workgroup per CU), 7 (two), 0 = chosen per launch (SV_SWEEP_CW)
    int n_cu = 256;
    // concurrent paths for small batches (run_paths_fanout)
    Buf Lbuf[8];
    int fanout_mode = 0;               // 0 auto, 1 always, -1 never
    int fan_chunk = 0;                 // frames per all-paths launch (SV_FAN_CHUNK); 0: from the geometry
    bool order_check = false;          // sv_debug_sweep_order(ctx, 1, ...): sweep workgroups stamp their start time
    Buf sw_starts;
    int sw_starts_n = 0, order_launches = 0, order_violations = 0;
    long long order_worst = 0, order_worst_prev = 0;
    int sweep_fault = 0;               // fault injection for tests (SV_SWEEP_FAULT): sweep launches left that report a timeout
    int last_agg_launches = 0, last_agg_sweeps = 0;  // of the most recent compute call (sv_last_aggregation)
    int last_sweep_cw = 0;             // compute waves per workgroup of the most recent sweep launch (sv_last_sweep_waves)
    int fuse_wta_min_batch = 6;  // batches below this run WTA as its own kernel (see run_sgm; crossover measured at 640x480 D=128)
    Buf disp_lr, out;        // i16 [H][W]
    Buf rgb_l, rgb_r;        // u8 [H][W][3] input staging
    Buf rect_l, rect_r;      // u8 [H][W][3] rectified
    Buf gray_l, gray_r;      // u8 [H][W]
    Buf maps[4];             // f32 [H][W]
    Buf mapper;              // f32 [H][W][2] staging for sv_depth_host with a caller-supplied mapper
    Buf mapper_lr[2];        // f32 [H][W][2] MapperL / MapperR made by sv_update_mapper
    bool have_mapper[2] = {false, false};
    Buf depth;               // f32 [H][W]
    Buf d16tmp;              // i16 [H][W] staging for host helper entry points
    Buf vis;                 // u8 [H][W]
    Buf lab, cnt;            // int [H][W] speckle union-find
    Buf wls_lut, wls_wh, wls_wv, wls_cur, wls_tmp;  // WLS: weight table, edge weights, work images (float)
    Buf wls_g, wls_v, wls_o16, wls_o8;              // WLS host entry point staging
    Buf world_cart, world_cyl;                      // f32 [H][W][3] staging of sv_world_image_host
    Buf blf_a, blf_b, blf_mm, blf_o16, blf_o8, blf_tab, blf_tmp, blf_gray;      // BLF: the two grids, min/max words, host entry point staging
    double blf_tab_sigma = -1.0;                    // colour / space tables on the device were built for these
    int blf_tab_radius = -1;
    float wls_lut_sigma = -1.0f;                    // sigma the device table was built for
    bool have_maps = false;  // both cameras
    bool have_map[2] = {false, false};

    // stage timing: a ring of event sets, one set per compute call, read back (and averaged)
    // only in sv_get_timings so that recording never synchronises the stream
    static constexpr int EV_RING = 64, EV_PER = SV_N_TIMINGS;  // 6 marks per call
    bool timing = false;
    hipEvent_t ev[EV_RING][EV_PER] = {};
    hipEvent_t ev_agg[EV_RING][8] = {};  // after each aggregation launch of the call (sv_get_aggregation_timings)
    int ev_agg_n[EV_RING] = {};
    bool ev_ok = false;
    long ev_calls = 0;  // compute calls recorded since sv_enable_timing
};

namespace {

const char* k_noctx = "null context";

int fail(sv_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail((c), SV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Geometry with min_disparity (EXT: cv::StereoSGBM as matchSGBM parameterises it, StereoMatcher.cpp:399; the CUDA
// back-end has no such parameter): the computable columns are [max(minD + D, 0), W + min(minD, 0)), cost index d
// pairs left x with right x - d - minD, invalid = (minD - 1) * 16.  minD = 0: columns [D, W).
inline int geo_col0(int W, const sv_params& p) { (void)W; return std::max(p.min_disparity + p.num_disparities, 0); }
inline int geo_colN(int W, const sv_params& p) { return W + std::min(p.min_disparity, 0); }
inline int geo_w1(int W, const sv_params& p) { return geo_colN(W, p) - geo_col0(W, p); }
inline int geo_inv(const sv_params& p) { return (p.min_disparity - 1) * 16; }

int check_params(sv_ctx* c, const sv_params* p, int W, int H) {
    if (!p) return fail(c, SV_ERR_INVALID_ARG, "params is NULL");
    const int D = p->num_disparities, BS = p->block_size;
    if (p->min_disparity < -1024 || p->min_disparity > 1024)
        return fail(c, SV_ERR_UNSUPPORTED, "min_disparity must be in [-1024, 1024] (got %d)", p->min_disparity);
    if (D < 32 || D > 256 || (D % 32) != 0)
        return fail(c, SV_ERR_UNSUPPORTED, "num_disparities must be a multiple of 32 in [32,256] (got %d)", D);
    if (BS < 1 || BS > 15 || (BS % 2) == 0)   // the module's GUI offers 3..31 (cvgui.cpp:100); even sizes: SURVEY A9b; beyond 15 the int16 test below fails for any cap
        return fail(c, SV_ERR_UNSUPPORTED, "block_size must be odd in [1,15] (got %d)", BS);
    // width: the LR check keeps one 4-byte key per column of a row in LDS (160 KB)
    if (D >= 32 && D <= 256 && (geo_w1(W, *p) < 1 || H < BS || geo_w1(W, *p) < BS || W > 40000 || H > 65535))
        return fail(c, SV_ERR_INVALID_ARG, "image %dx%d too small/large for D=%d min_disparity=%d block=%d (W <= 40000, H <= 65535)", W, H, D,
                    p->min_disparity, BS);
    // clip(v) + cap lives in a byte (PixType, costs.cu:286-292): 2 * cap <= 255.  The GUI offers 0..100 (cvgui.cpp:102).
    if (p->pre_filter_cap < 0 || p->pre_filter_cap > 127)
        return fail(c, SV_ERR_UNSUPPORTED, "pre_filter_cap must be in [0,127] (got %d)", p->pre_filter_cap);
    if (p->P1 < 0 || p->P2 < p->P1 || p->P2 < 1) return fail(c, SV_ERR_INVALID_ARG, "need 0 <= P1 <= P2, P2 >= 1");
    const int cn = (p->flags & SV_FLAG_COST_RGB) ? 3 : 1;
    const long cmax = (long)cn * (2 * p->pre_filter_cap + 63) * BS * BS + p->P2;
    if (cmax > 32767 || (long)p->P1 + 32767 > 65535)
        return fail(c, SV_ERR_UNSUPPORTED, "cost range overflows int16 (max C = %d x (2 x cap + 63) x block^2 + P2 = %ld > 32767; the reference would wrap silently): reduce block_size, pre_filter_cap or P2", cn, cmax);
    if (p->uniqueness_ratio < 0 || p->uniqueness_ratio > 100)
        return fail(c, SV_ERR_INVALID_ARG, "uniqueness_ratio must be in [0,100]");
    if ((p->path_mask & 0xFFu) == 0 || (p->path_mask & ~0xFFu)) return fail(c, SV_ERR_INVALID_ARG, "bad path_mask 0x%x", p->path_mask);
    if (p->speckle_window_size < 0 || p->speckle_range < 0) return fail(c, SV_ERR_INVALID_ARG, "negative speckle parameter");
    return SV_OK;
}

int ensure(sv_ctx* c, Buf& b, size_t bytes) {
    if (b.n >= bytes && b.p) return SV_OK;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.n = 0; }
    const hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the context stays usable (a smaller sv_set_batch may follow)
        b.p = nullptr;
        b.n = 0;
        return fail(c, SV_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    b.n = bytes;
    return SV_OK;
}
void release(Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.n = 0;
}

// sizes of the core buffers for a batch of nb frames with the current parameters
struct CoreSizes { size_t stat_l, stat_r, vol, rec, plane; };
CoreSizes core_sizes(const sv_ctx* c, int nb) {
    const size_t npx = (size_t)c->W * c->H * nb;
    const size_t nvol = (size_t)geo_w1(c->W, c->prm) * c->H * c->prm.num_disparities * nb;
    const size_t cn = (c->prm.flags & SV_FLAG_COST_RGB) ? 3 : 1;
    return CoreSizes{npx * 8 * cn, (size_t)6 * c->H * svk::cost_plane_pitch(c->W) * 2 * nb * cn, nvol * 2, npx * 8, npx * 2};
}

int alloc_core_once(sv_ctx* c) {
    const size_t npx = (size_t)c->W * c->H * c->batch;
    const size_t nvol = (size_t)geo_w1(c->W, c->prm) * c->H * c->prm.num_disparities * c->batch;
    int rc;
    const size_t cn = (c->prm.flags & SV_FLAG_COST_RGB) ? 3 : 1;  // statistics per colour plane
    if ((rc = ensure(c, c->stat_l, npx * 8 * cn))) return rc;
    // right statistics: three u16 planes (two 8-bit statistics each) x two copies (k_stats_planes)
    if ((rc = ensure(c, c->stat_r, (size_t)6 * c->H * svk::cost_plane_pitch(c->W) * 2 * c->batch * cn))) return rc;
    if ((rc = ensure(c, c->C, nvol * 2))) return rc;
    if ((rc = ensure(c, c->S, nvol * 2))) return rc;
    if ((rc = ensure(c, c->rec, npx * 8))) return rc;
    if ((rc = ensure(c, c->disp_lr, npx * 2))) return rc;
    if ((rc = ensure(c, c->out, npx * 2))) return rc;
    return SV_OK;
}

// (Re)allocates the core buffers for c->batch / c->prm.  ensure() frees a buffer before it asks for
// the larger one (two 70 GB volumes cannot both be held twice), so a failed allocation leaves holes:
// then every core buffer is released and the geometry the context had before (old_batch, old_prm:
// buffers it did hold a moment ago) is allocated again.  If even that fails the context is left
// with core_ok = false and every compute call is rejected with SV_ERR_NOMEM instead of launching
// kernels on NULL volumes.
int alloc_core(sv_ctx* c, int old_batch, const sv_params& old_prm) {
    int rc = alloc_core_once(c);
    if (rc == SV_OK) { c->core_ok = true; return SV_OK; }
    char why[sizeof(c->err)];
    snprintf(why, sizeof(why), "%s", c->err);
    Buf* core[] = {&c->stat_l, &c->stat_r, &c->C, &c->S, &c->rec, &c->disp_lr, &c->out};
    for (Buf* b : core) release(*b);
    const int want_batch = c->batch;
    c->batch = old_batch;
    c->prm = old_prm;
    c->core_ok = alloc_core_once(c) == SV_OK;
    if (!c->core_ok)
        for (Buf* b : core) release(*b);
    fail(c, rc, "%s; batch %d not available, context %s", why, want_batch,
         c->core_ok ? "keeps its previous batch size and parameters" : "has no volumes now: call sv_set_batch with a smaller size");
    return rc;
}

// Strided host<->device image copy.  Dense rows (the usual case: cv::Mat::isContinuous) go through
// one linear copy: hipMemcpy2DAsync on pageable host memory costs milliseconds per 640x480 image.
hipError_t copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                  hipStream_t stream) {
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * height, kind, stream);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream);
}

struct DevGuard {
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess) ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
        else ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <int N>
void launch_agg_n(sv_ctx* c, int dir, int mode, int nb, uint32_t* Sout, hipStream_t st) {
    static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    const int W1 = geo_w1(c->W, c->prm), H = c->H;
    const int dx = DX[dir], dy = DY[dir];
    const bool vert = dy != 0;
    const int nchains = vert ? W1 : H;
    const dim3 grid((nchains + 3) / 4, nb), block(64);
    const uint32_t P1P1 = (uint32_t)c->prm.P1 * 0x10001u, P2P2 = (uint32_t)c->prm.P2 * 0x10001u;
    const uint32_t* Cv = (const uint32_t*)c->C.p;
    uint32_t* Sv = Sout;
    constexpr int U = N <= 4 ? 4 : 2;
    uint2* rec = (uint2*)c->rec.p;
    const int uq = c->prm.uniqueness_ratio;
#define SV_LAUNCH_AGG(V, M) \
    hipLaunchKernelGGL((svk::k_agg<N, V, M, U>), grid, block, 0, st, Cv, Sv, W1, H, dx, dy, P1P1, P2P2, rec, uq, c->W, \
                       geo_col0(c->W, c->prm), c->prm.min_disparity)
    if (vert) {
        switch (mode) {
            case 0: SV_LAUNCH_AGG(true, 0); break;
            case 1: SV_LAUNCH_AGG(true, 1); break;
            case 2: SV_LAUNCH_AGG(true, 2); break;
            default: SV_LAUNCH_AGG(true, 3); break;
        }
    } else {
        switch (mode) {
            case 0: SV_LAUNCH_AGG(false, 0); break;
            case 1: SV_LAUNCH_AGG(false, 1); break;
            case 2: SV_LAUNCH_AGG(false, 2); break;
            default: SV_LAUNCH_AGG(false, 3); break;
        }
    }
#undef SV_LAUNCH_AGG
}

void launch_agg_to(sv_ctx* c, int dir, int mode, int nb, uint32_t* Sout, hipStream_t st) {
    switch (c->prm.num_disparities / 32) {
        case 1: launch_agg_n<1>(c, dir, mode, nb, Sout, st); break;
        case 2: launch_agg_n<2>(c, dir, mode, nb, Sout, st); break;
        case 3: launch_agg_n<3>(c, dir, mode, nb, Sout, st); break;
        case 4: launch_agg_n<4>(c, dir, mode, nb, Sout, st); break;
        case 5: launch_agg_n<5>(c, dir, mode, nb, Sout, st); break;
        case 6: launch_agg_n<6>(c, dir, mode, nb, Sout, st); break;
        case 7: launch_agg_n<7>(c, dir, mode, nb, Sout, st); break;
        default: launch_agg_n<8>(c, dir, mode, nb, Sout, st); break;
    }
}
void launch_agg(sv_ctx* c, int dir, int mode, int nb) { launch_agg_to(c, dir, mode, nb, (uint32_t*)c->S.p, c->stream); }

// frames f0 .. f0 + nb - 1 of the batch (the L volumes hold one chunk and are reused by the next)
template <int N>
void launch_paths_sum_n(sv_ctx* c, const svk::PathSet& ps, const svk::PathBufs& pb, int f0, int nb) {
    const int W1 = geo_w1(c->W, c->prm), H = c->H;
    constexpr int U = N <= 4 ? 4 : 2;
    const int nch = W1 > H ? W1 : H;
    const size_t npx = (size_t)W1 * H;
    const uint32_t* Cf = (const uint32_t*)c->C.p + (size_t)f0 * npx * (16 * N);
    uint2* recf = (uint2*)c->rec.p + (size_t)f0 * c->H * c->W;
    hipLaunchKernelGGL((svk::k_agg_paths<N, U>), dim3((nch + 3) / 4, nb, pb.n), dim3(64), 0, c->stream, Cf, ps,
                       W1, H, (uint32_t)c->prm.P1 * 0x10001u, (uint32_t)c->prm.P2 * 0x10001u);
    hipLaunchKernelGGL((svk::k_sum_wta<N>), dim3((unsigned)((npx * 16 + 255) / 256), nb), dim3(256), 0, c->stream, pb,
                       recf, c->W, c->H, c->prm.uniqueness_ratio, geo_col0(c->W, c->prm), W1, c->prm.min_disparity);
}

// Small batches are latency-bound (one path launch of one 640x480 frame is 128 waves on 1024
// SIMDs), so all paths run side by side in one launch, each writing its own L volume, then one
// streaming kernel sums them and runs WTA.  More bytes (3V per path instead of ~2.75V) but the
// eight recurrences overlap.
int run_paths_fanout(sv_ctx* c, int f_start, int nb, int chunk) {
    static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    const sv_params& p = c->prm;
    if (chunk < 1 || chunk > nb) chunk = nb;
    const size_t vol = (size_t)geo_w1(c->W, p) * c->H * p.num_disparities * 2 * chunk;
    svk::PathSet ps;
    svk::PathBufs pb;
    pb.n = 0;
    for (int dir = 0; dir < 8; dir++)
        if (p.path_mask & (1u << dir)) {
            const int i = pb.n++;
            int rc = ensure(c, c->Lbuf[i], vol);
            if (rc) return rc;
            ps.L[i] = (uint32_t*)c->Lbuf[i].p;
            ps.dx[i] = (signed char)DX[dir];
            ps.dy[i] = (signed char)DY[dir];
            pb.L[i] = (const uint32_t*)c->Lbuf[i].p;
        }
    for (int i = pb.n; i < 8; i++) { ps.L[i] = ps.L[0]; ps.dx[i] = ps.dx[0]; ps.dy[i] = ps.dy[0]; pb.L[i] = pb.L[0]; }
    for (int f0 = f_start; f0 < f_start + nb; f0 += chunk) {
        const int n = f_start + nb - f0 < chunk ? f_start + nb - f0 : chunk;
        c->last_agg_launches += 1;
        switch (p.num_disparities / 32) {
            case 1: launch_paths_sum_n<1>(c, ps, pb, f0, n); break;
            case 2: launch_paths_sum_n<2>(c, ps, pb, f0, n); break;
            case 3: launch_paths_sum_n<3>(c, ps, pb, f0, n); break;
            case 4: launch_paths_sum_n<4>(c, ps, pb, f0, n); break;
            case 5: launch_paths_sum_n<5>(c, ps, pb, f0, n); break;
            case 6: launch_paths_sum_n<6>(c, ps, pb, f0, n); break;
            case 7: launch_paths_sum_n<7>(c, ps, pb, f0, n); break;
            default: launch_paths_sum_n<8>(c, ps, pb, f0, n); break;
        }
    }
    return SV_OK;
}

// dpart0, nparts: the launch costs disparities dpart0 .. dpart0 + 32 N nparts - 1 of the volume's D (k_cost2: disparity parts)
template <int N, int CN, int NT = 512>
int launch_cost2_n(sv_ctx* c, int nb, int dpart0 = 0, int nparts = 1, bool col0 = true) {
    // CN = colour planes in the cost: 1 (gray) or 3 (SV_FLAG_COST_RGB)
    constexpr int NCOL = NT / 16;
    const sv_params& p = c->prm;
    const int W = c->W, H = c->H, D = p.num_disparities, W1 = geo_w1(W, p), r = p.block_size / 2;
    const int TX = NCOL - 2 * r, WP = svk::cost_plane_pitch(W);
    if constexpr (NT == 512 && CN == 1 && N <= 4) {
        // Large batches: 64-column tiles (1024 threads, one workgroup per CU) recompute the 2r halo columns of the
        // BT costs for 58 instead of 26 output columns: 27.9 -> 27.2 ms per 1024 frames at C2.  Only where the
        // tile fits the 160 KB of LDS (D <= 128 with blocks up to 7x7) and the launch has enough tiles to go round.
        if (nb >= 8 && 64 - 2 * r >= 16 && svk::cost_lds_bytes<N, 1024, CN>(r) <= 160 * 1024 && c->debug_mode != 4 && !getenv("SV_COST_NT512"))
            return launch_cost2_n<N, CN, 1024>(c, nb, dpart0, nparts, col0);
    }
    if (TX < 1) return fail(c, SV_ERR_UNSUPPORTED, "block size %d leaves no output column in a %d-column tile", p.block_size, NCOL);
    const size_t lds = svk::cost_lds_bytes<N, NT, CN>(r);
    if (lds > 160 * 1024) {
        // D >= 224 with blocks from 11 x 11: the ring of a 32-column tile passes the 160 KB.  24 tile columns fit.
        if constexpr (NT == 512 && N >= 7) return launch_cost2_n<N, CN, 384>(c, nb, dpart0, nparts, col0);
        return fail(c, SV_ERR_UNSUPPORTED, "cost kernel tile of %zu bytes exceeds the 160 KB of LDS", lds);
    }
    bool* attr_set = c->attr_cost2[CN == 3][NT == 512 ? 0 : (NT == 384 ? 1 : 2)];  // per context: the attribute is per device
    if (lds > 48 * 1024 && !attr_set[N - 1]) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&svk::k_cost2<N, NT, CN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        attr_set[N - 1] = true;
    }
    const int ntx = (W1 + TX - 1) / TX;
    // bands of rows: every band recomputes a 2r-row prologue, so as few bands as still give each CU
    // eight workgroups over the launch (a large batch needs none: one band per column tile)
    int bands = 1;
    while ((size_t)ntx * bands * nb < (size_t)8 * c->n_cu && (H + bands) / (bands + 1) >= 24) bands++;
    const int yband = (H + bands - 1) / bands;
    const int nbands = (H + yband - 1) / yband;
    const bool by_xcd = nb >= 8;  // below that the frame -> XCD order would leave XCDs idle
    const size_t n_wg = (size_t)ntx * nbands * nparts * (by_xcd ? (nb + 7) / 8 * 8 : nb);
    if (n_wg > 0x7FFFFFFFu) return fail(c, SV_ERR_UNSUPPORTED, "batch too large for one cost launch");
    if (int rc = ensure(c, c->cost_sink, (size_t)svk::COST_SINK_SLOTS * NT * N * 4)) return rc;
    hipLaunchKernelGGL((svk::k_cost2<N, NT, CN>), dim3((unsigned)n_wg), dim3(NT), lds, c->stream,
                       (const uint2*)c->stat_l.p, (const uint16_t*)c->stat_r.p, (uint32_t*)c->C.p, W, H, WP, r, yband,
                       (uint32_t)p.P2 * 0x10001u, (p.flags & SV_FLAG_C_TAIL_P2) ? 1 : 0, ntx, nbands, by_xcd ? nb : -nb,
                       geo_col0(W, p), W1, p.min_disparity, D / 2, dpart0, nparts, (uint32_t*)c->cost_sink.p);
    if (H > 1 && col0)
        hipLaunchKernelGGL(svk::k_cost_col0, dim3(H - 1, nb), dim3(128), 0, c->stream, (uint32_t*)c->C.p, W1, H, D / 2,
                           (uint32_t)p.P2 * 0x10001u, (p.flags & SV_FLAG_C_COL0_P2) ? 1 : 0);
    return SV_OK;
}
int launch_cost2(sv_ctx* c, int nb) {
    const bool rgb = (c->prm.flags & SV_FLAG_COST_RGB) != 0;
    // D > 128, gray, large batches: two disparity parts of at most 128 on the D <= 128 kernel's 64-column tiles and LDS stage
    // (C4, 98 frames: 39.6 ms as one N = 8 part, two waves per SIMD on 32-column tiles)
    const int n32 = c->prm.num_disparities / 32, r = c->prm.block_size / 2;
    if (!rgb && n32 > 4 && nb >= 8 && 64 - 2 * r >= 16 && svk::cost_lds_bytes<4, 1024, 1>(r) <= 160 * 1024 && c->debug_mode != 4 &&
        !getenv("SV_COST_NOPARTS")) {
        switch (n32) {
            case 5: if (int rc = launch_cost2_n<3, 1>(c, nb, 0, 1, false)) return rc; return launch_cost2_n<2, 1>(c, nb, 96, 1);
            case 6: return launch_cost2_n<3, 1>(c, nb, 0, 2);
            case 7: if (int rc = launch_cost2_n<4, 1>(c, nb, 0, 1, false)) return rc; return launch_cost2_n<3, 1>(c, nb, 128, 1);
            default: return launch_cost2_n<4, 1>(c, nb, 0, 2);
        }
    }
    switch (c->prm.num_disparities / 32) {
        case 1: return rgb ? launch_cost2_n<1, 3>(c, nb) : launch_cost2_n<1, 1>(c, nb);
        case 2: return rgb ? launch_cost2_n<2, 3>(c, nb) : launch_cost2_n<2, 1>(c, nb);
        case 3: return rgb ? launch_cost2_n<3, 3>(c, nb) : launch_cost2_n<3, 1>(c, nb);
        case 4: return rgb ? launch_cost2_n<4, 3>(c, nb) : launch_cost2_n<4, 1>(c, nb);
        case 5: return rgb ? launch_cost2_n<5, 3>(c, nb) : launch_cost2_n<5, 1>(c, nb);
        case 6: return rgb ? launch_cost2_n<6, 3>(c, nb) : launch_cost2_n<6, 1>(c, nb);
        case 7: return rgb ? launch_cost2_n<7, 3>(c, nb) : launch_cost2_n<7, 1>(c, nb);
        default: return rgb ? launch_cost2_n<8, 3>(c, nb) : launch_cost2_n<8, 1>(c, nb);
    }
}
template <int N>
void launch_wta_n(sv_ctx* c, int nb) {
    const size_t npx = (size_t)c->W * c->H;
    const dim3 grid((unsigned)((npx * 16 + 255) / 256), nb), block(256);
    hipLaunchKernelGGL((svk::k_wta<N>), grid, block, 0, c->stream, (const uint32_t*)c->S.p, (uint2*)c->rec.p, c->W, c->H,
                       c->prm.uniqueness_ratio, geo_col0(c->W, c->prm), geo_w1(c->W, c->prm), c->prm.min_disparity);
}
void launch_wta(sv_ctx* c, int nb) {
    switch (c->prm.num_disparities / 32) {
        case 1: launch_wta_n<1>(c, nb); break;
        case 2: launch_wta_n<2>(c, nb); break;
        case 3: launch_wta_n<3>(c, nb); break;
        case 4: launch_wta_n<4>(c, nb); break;
        case 5: launch_wta_n<5>(c, nb); break;
        case 6: launch_wta_n<6>(c, nb); break;
        case 7: launch_wta_n<7>(c, nb); break;
        default: launch_wta_n<8>(c, nb); break;
    }
}

// One fused sweep (four paths) over nb frames.  second = false: paths R, DR, D, DL, S is written;
// second = true: paths L, UL, U, UR on mirrored coordinates, S is read, WTA records are written.
// Dispatch-order check (sv_debug_sweep_order): reads back the start stamps of the last checked sweep launch.  The rule
// the oversubscribed case rests on (sv_sweep.hip.h): workgroup i goes to XCD i mod 8 and every XCD starts ITS workgroups
// in blockIdx order.  Counted: workgroups that started more than SW_ORDER_TOL ticks (100 MHz) before workgroup i - 8,
// their predecessor on the same XCD.  (Against workgroup i - 1, which sits on another XCD, leads of 100+ us are normal
// in an oversubscribed launch: measured, tests/test_gpu_sweep.py.)
constexpr long long SW_ORDER_TOL = 2000;   // 20 us: workgroups dispatched together start within microseconds of each other
constexpr int SW_N_XCD = 8;                // gfx950: 8 XCDs of 32 CUs
int sweep_order_collect(sv_ctx* c) {
    if (c->sw_starts_n <= 0 || !c->sw_starts.p) return SV_OK;
    std::vector<unsigned long long> st((size_t)c->sw_starts_n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(st.data(), c->sw_starts.p, st.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 1; i < st.size(); i++) {
        const long long lead1 = (long long)st[i - 1] - (long long)st[i];   // > 0: i started before i - 1 (another XCD)
        if (lead1 > c->order_worst_prev) c->order_worst_prev = lead1;
        if (i < (size_t)SW_N_XCD) continue;
        const long long lead = (long long)st[i - SW_N_XCD] - (long long)st[i];   // > 0: i started before its XCD's previous workgroup
        if (lead > SW_ORDER_TOL) c->order_violations++;
        if (lead > c->order_worst) c->order_worst = lead;
    }
    c->order_launches++;
    c->sw_starts_n = 0;
    return SV_OK;
}

template <int N, int CW>
int launch_sweep_n(sv_ctx* c, bool second, int nb) {
    constexpr int ROWS = 4 * CW, NTHR = (CW + 1) * 64;
    const int D = c->prm.num_disparities, W1 = geo_w1(c->W, c->prm), H = c->H;
    const int n_bands = (H + ROWS - 1) / ROWS;
    const size_t lds = second ? svk::sweep_lds_bytes<N, CW, true>() : svk::sweep_lds_bytes<N, CW, false>();
    const void* fn = second ? reinterpret_cast<const void*>(&svk::k_sweep<N, true, CW>) : reinterpret_cast<const void*>(&svk::k_sweep<N, false, CW>);
    int& occ = c->occ_sweep[second][N - 1][CW == 15];  // [2][8][2], per context: workgroups of this kernel a CU holds (0: not asked yet)
    if (occ == 0) {
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, NTHR, lds) != hipSuccess || q < 1) { (void)hipGetLastError(); q = -1; }
        occ = q;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2048 / NTHR) per_cu = 2048 / NTHR;
    if (occ > 0 && occ < per_cu) per_cu = occ;   // what the runtime says fits (registers included), when it is less than LDS and threads allow
    if (per_cu < 1) per_cu = 1;
    // Frame pipelines (groups of n_bands workgroups) the chip holds at once.  A launch never has more: band b waits for
    // band b-1 of its group, and only a launch whose workgroups are all resident is independent of the order in which
    // the dispatcher starts them.  More frames than that = more frames per group, and when a group's
    // block of volumes would pass the kernel's 32-bit offsets, further launches.  (SV_SWEEP_NCU overrides the CU count
    // for tests, which oversubscribe on purpose.)
    int resident = c->n_cu * per_cu / n_bands;
    if (resident < 1) resident = 1;
    int n_groups = resident < nb ? resident : nb;
    int fpg = (nb + n_groups - 1) / n_groups;
    // a group's block of C / S must stay below 2^32 dwords (the kernel keeps 32-bit dword offsets from the
    // group's first frame and widens them when it forms the address)
    const size_t frame_bytes = (size_t)H * W1 * D * 2;
    const int fpg_max = (int)(((size_t)1 << 34) / frame_bytes) - 1;
    if (fpg_max < 1) return fail(c, SV_ERR_UNSUPPORTED, "frame too large for the sweep kernel");
    if (fpg > fpg_max) fpg = fpg_max;
    if ((long)fpg * W1 > (1L << 30)) return fail(c, SV_ERR_UNSUPPORTED, "batch too large for the sweep kernel");
    const int ho_px = (fpg * W1 + 15) & ~15;
    constexpr int REC = svk::sweep_rec_words<N>();
    const int frames_per_launch = resident * fpg;   // >= nb unless fpg was clamped
    int rc;
    for (int f0 = 0; f0 < nb; f0 += frames_per_launch) {
        const int nf = nb - f0 < frames_per_launch ? nb - f0 : frames_per_launch;
        n_groups = (nf + fpg - 1) / fpg;  // contiguous blocks of fpg frames; the last block may be shorter
        const int grid = n_groups * n_bands;
        if ((rc = ensure(c, c->sw_handoff, (size_t)n_groups * (n_bands > 1 ? n_bands - 1 : 1) * (size_t)(ho_px + 1) * REC * 4))) return rc;
        if ((rc = ensure(c, c->sw_progress, (size_t)n_groups * n_bands * 4))) return rc;
        if ((rc = ensure(c, c->sw_sink, (size_t)grid * NTHR * svk::sweep_sink_words<N>() * 4))) return rc;
        HIPCHK(c, hipMemsetAsync(c->sw_progress.p, 0, (size_t)n_groups * n_bands * 4, c->stream));
        svk::SweepArgs a;
        a.C = (const uint32_t*)c->C.p + (size_t)f0 * (frame_bytes / 4);
        a.S = (uint32_t*)c->S.p + (size_t)f0 * (frame_bytes / 4);
        a.rec = (uint2*)c->rec.p + (size_t)f0 * H * c->W;
        a.handoff = (unsigned long long*)c->sw_handoff.p;
        a.progress = (unsigned int*)c->sw_progress.p;
        a.error = c->sw_error;
        a.sink = (uint32_t*)c->sw_sink.p;
        a.W1 = W1; a.H = H; a.W = c->W;
        a.col0 = geo_col0(c->W, c->prm); a.min_disp = c->prm.min_disparity;
        a.P1P1 = (uint32_t)c->prm.P1 * 0x10001u;
        a.P2P2 = (uint32_t)c->prm.P2 * 0x10001u;
        a.uniq = c->prm.uniqueness_ratio;
        a.n_frames = nf; a.n_groups = n_groups; a.n_bands = n_bands; a.ho_px = ho_px; a.fpg = fpg;
        a.starts = nullptr;
        if (c->order_check) {
            if ((rc = ensure(c, c->sw_starts, (size_t)grid * 8))) return rc;
            // the previous launch's stamps (same stream, finished by now) are judged before they are overwritten
            if ((rc = sweep_order_collect(c))) return rc;
            a.starts = (unsigned long long*)c->sw_starts.p;
            c->sw_starts_n = grid;
        }
        a.prof = nullptr;
#ifdef SV_SWEEP_PROF
        if ((rc = ensure(c, c->sw_prof, (size_t)grid * (CW + 1) * 4 * 8 * 2))) return rc;
        a.prof = (unsigned long long*)c->sw_prof.p + (second ? (size_t)grid * (CW + 1) * 4 : 0);
        c->sw_prof_n = grid * (CW + 1) * 4 * 2;
#endif
        if (second) hipLaunchKernelGGL((svk::k_sweep<N, true, CW>), dim3(grid), dim3(NTHR), lds, c->stream, a);
        else hipLaunchKernelGGL((svk::k_sweep<N, false, CW>), dim3(grid), dim3(NTHR), lds, c->stream, a);
    }
    return SV_OK;
}
template <int CW>
int launch_sweep_cw(sv_ctx* c, bool second, int nb) {
    if constexpr (CW == 8) {
        // D > 128: 8 compute waves (two per SIMD) + the service wave, 32 rows per band; a lane keeps up to 132 VGPRs, and
        // 32 rows of 1.6 KB records + the winner-take-all slots are what the 160 KB of LDS hold (sweep_ring)
        switch (c->prm.num_disparities / 32) {
            case 5: return launch_sweep_n<5, CW>(c, second, nb);
            case 6: return launch_sweep_n<6, CW>(c, second, nb);
            case 7: return launch_sweep_n<7, CW>(c, second, nb);
            case 8: return launch_sweep_n<8, CW>(c, second, nb);
            default: break;
        }
    } else {
        switch (c->prm.num_disparities / 32) {
            case 1: return launch_sweep_n<1, CW>(c, second, nb);
            case 2: return launch_sweep_n<2, CW>(c, second, nb);
            case 3: return launch_sweep_n<3, CW>(c, second, nb);
            case 4: return launch_sweep_n<4, CW>(c, second, nb);
            default: break;
        }
    }
    return fail(c, SV_ERR_UNSUPPORTED, "fused sweeps need D <= 256");
}
// Frame pipelines a sweep launch can keep resident: workgroup slots / bands per frame.
int sweep_pipes(const sv_ctx* c, int cw) {
    const int rows = 4 * cw, bands = (c->H + rows - 1) / rows;
    const int per_cu = (cw == 15 || c->prm.num_disparities > 128) ? 1 : 2;
    const int pipes = c->n_cu * per_cu / bands;
    return pipes > 0 ? pipes : 1;
}
// Compute waves per sweep workgroup for a launch of nb frames.  D > 128 needs the 8-wave form (VGPRs).
// Otherwise the form with the shorter predicted launch: a launch streams ceil(nb / pipelines) frames
// of W1 steps through every pipeline after (bands - 1) * (2 * rows + 2) steps of fill.  Measured at
// 640x480 D=128 (scratch/perf_cw.py), in units of a 16-wave streaming step: 16-wave workgroups 1.0 per
// streaming step and 1.44 per fill step, 8-wave workgroups (two per CU) 1.063 and 0.83.
// predicted length of one sweep launch over n frames with cw compute waves per workgroup, in 16-wave streaming steps
double sweep_cost(const sv_ctx* c, int cw, int n) {
    const int W1 = geo_w1(c->W, c->prm);
    const int rows = 4 * cw, bands = (c->H + rows - 1) / rows, pipes = sweep_pipes(c, cw);
    const double stream = cw == 15 ? 1.0 : 1.063, fill = cw == 15 ? 1.44 : 0.83;
    return (double)((n + pipes - 1) / pipes) * W1 * stream + (double)(bands - 1) * (2 * rows + 2) * fill;
}
int sweep_cw_for(const sv_ctx* c, int nb) {
    if (c->prm.num_disparities > 128) return 8;
    if (c->sweep_cw) return c->sweep_cw;
    return sweep_cost(c, 7, nb) < sweep_cost(c, 15, nb) ? 7 : 15;
}
// How many of nb frames go through the sweeps when the rest may take the all-paths kernels (run_paths_fanout): a launch
// streams whole rounds of one frame per pipeline, so a few frames beyond a full round cost a whole round more (32 frames
// on 28 pipelines: 6.5 ms against 4.8 ms for 28) - unless they go the other way.  One frame through the all-paths
// kernels costs about 3.1e-4 streaming steps per pixel (640x480, D=128: 0.26 ms = 76 steps of the two sweeps).
int sweep_share(const sv_ctx* c, int nb) {
    const int W1 = geo_w1(c->W, c->prm);
    const double fan_unit = 3.1e-4 * W1 * c->H;
    int best_n = nb;
    double best = sweep_cost(c, sweep_cw_for(c, nb), nb);
    for (int cw = 7; cw <= 15; cw++) {
        if (cw != 7 && cw != 8 && cw != 15) continue;
        if ((c->prm.num_disparities > 128) != (cw == 8) || (c->sweep_cw && cw != 8 && cw != c->sweep_cw)) continue;
        const int pipes = sweep_pipes(c, cw), n = nb / pipes * pipes;
        if (n <= 0 || n == nb) continue;
        const double cost = sweep_cost(c, cw, n) + (nb - n) * fan_unit;
        if (cost < best) { best = cost; best_n = n; }
    }
    return best_n;
}
int launch_sweep(sv_ctx* c, bool second, int nb) {
    c->last_sweep_cw = sweep_cw_for(c, nb);
    const int rc = c->last_sweep_cw == 15 ? launch_sweep_cw<15>(c, second, nb)
                   : (c->last_sweep_cw == 8 ? launch_sweep_cw<8>(c, second, nb) : launch_sweep_cw<7>(c, second, nb));
    if (rc == SV_OK && second && c->sweep_fault > 0 && c->sw_error) {   // what a band that ran out its bounded wait leaves behind
        c->sweep_fault--;
        *c->sw_error = 1;
    }
    return rc;
}
// a sweep kernel gave up waiting for its neighbour band (never expected; reported at the next sync point)
int check_sweep_error(sv_ctx* c) {
    if (c->sw_error && *c->sw_error) {
        *c->sw_error = 0;
        // the sweeps rest on workgroups starting in blockIdx order whenever a launch is not fully resident (sv_sweep.hip.h);
        // a context that has seen them fail once stays on the per-path kernels, which have no such dependence
        c->sweep_mode = -1;
        return fail(c, SV_ERR_HIP, "fused sweep kernel timed out waiting for a neighbouring band; results of that call are invalid "
                                   "(this context uses the per-path kernels from now on: repeat the call)");
    }
    return SV_OK;
}

// cv::filterSpeckles on nb dense-packed frames, in place.  lab / cnt: one int per pixel each (labels are
// pixel indices within the frame, so a frame must stay below 2^31 pixels; frames are a grid dimension).
int speckle_launch(sv_ctx* c, int16_t* img, int* lab, int* cnt, int W, int H, int max_size, int max_diff, int nb, hipStream_t st) {
    const int new_val = geo_inv(c->prm);   // cv::filterSpeckles(disp, (minDisparity - 1) * 16, ...) inside cv::StereoSGBM::compute
    if ((size_t)W * H >= ((size_t)1 << 31)) return fail(c, SV_ERR_UNSUPPORTED, "speckle filter: frame of %dx%d pixels exceeds 2^31 labels", W, H);
    if (nb < 1 || nb > 65535) return fail(c, SV_ERR_INVALID_ARG, "speckle filter: %d frames per call (1..65535)", nb);
    const int n = W * H;
    const dim3 b(256), g1((n + 255) / 256, nb), g2((W + 255) / 256, H, nb);
    hipLaunchKernelGGL(svk::k_speckle_init, g1, b, 0, st, lab, cnt, n);
    hipLaunchKernelGGL(svk::k_speckle_merge, g2, b, 0, st, img, lab, W, H, new_val, max_diff);
    hipLaunchKernelGGL(svk::k_speckle_count, dim3((n + 1023) / 1024, nb), dim3(1024), 0, st, img, lab, cnt, n, new_val);
    hipLaunchKernelGGL(svk::k_speckle_apply, g1, b, 0, st, img, lab, cnt, n, new_val, max_size);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}
int speckle_dev(sv_ctx* c, int16_t* img, int max_size, int max_diff, int nb) {
    const size_t n = (size_t)c->W * c->H * nb;
    int rc;
    if ((rc = ensure(c, c->lab, n * 4))) return rc;
    if ((rc = ensure(c, c->cnt, n * 4))) return rc;
    return speckle_launch(c, img, (int*)c->lab.p, (int*)c->cnt.p, c->W, c->H, max_size, max_diff, nb, c->stream);
}

int inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], cc = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + cc * C;
    if (det == 0.0) return -1;
    const double id = 1.0 / det;
    o[0] = A * id; o[1] = -(b * i - cc * h) * id; o[2] = (b * f - cc * e) * id;
    o[3] = B * id; o[4] = (a * i - cc * g) * id;  o[5] = -(a * f - cc * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id;  o[8] = (a * e - b * d) * id;
    return 0;
}

// K 3x3, dist[n_dist <= 8], R 3x3, P 3x4 -> kernel argument; inverse = true for initUndistortRectifyMap
int cam_model(sv_ctx* c, const double* K, const double* dist, int n_dist, const double* R, const double* P, bool inverse,
              svk::CamModel* out) {
    if (!K || !R || !P || (n_dist > 0 && !dist)) return fail(c, SV_ERR_INVALID_ARG, "NULL camera matrix");
    if (n_dist < 0 || n_dist > 8) return fail(c, SV_ERR_UNSUPPORTED, "distortion model with %d coefficients (0..8 supported: k1 k2 p1 p2 k3 k4 k5 k6)", n_dist);
    if (K[0] == 0.0 || K[4] == 0.0) return fail(c, SV_ERR_INVALID_ARG, "zero focal length");
    const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double PR[9];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) PR[3 * r + q] = P3[3 * r] * R[q] + P3[3 * r + 1] * R[3 + q] + P3[3 * r + 2] * R[6 + q];
    if (inverse) {
        if (inv3(PR, out->m)) return fail(c, SV_ERR_INVALID_ARG, "P*R is singular");
    } else {
        for (int i = 0; i < 9; i++) out->m[i] = PR[i];
    }
    out->fx = K[0]; out->fy = K[4]; out->cx = K[2]; out->cy = K[5];
    for (int i = 0; i < 8; i++) out->k[i] = i < n_dist ? dist[i] : 0.0;
    return SV_OK;
}

// computeROI2 (StereoMatcher.cpp:12-29)
void wls_roi(const sv_ctx* c, int roi[4]) {
    const int bs2 = c->prm.block_size / 2;
    const int minD = c->prm.min_disparity, maxD = minD + c->prm.num_disparities - 1;
    roi[0] = maxD + bs2;
    roi[1] = bs2;
    roi[2] = (c->W + minD - bs2) - roi[0];
    roi[3] = (c->H - bs2) - roi[1];
}

// filterWLS (StereoMatcher.cpp:154-160) on nf dense-packed frames; vis/out8 may be NULL.
int wls_dev(sv_ctx* c, int nf, const int16_t* d16, const uint8_t* vis, const uint8_t* guide, int16_t* out16, uint8_t* out8) {
    const int W = c->W, H = c->H;
    int roi[4];
    wls_roi(c, roi);
    const int rx = roi[0], ry = roi[1], rw = roi[2], rh = roi[3];
    if (rw < 1 || rh < 1) return fail(c, SV_ERR_INVALID_ARG, "empty WLS ROI %dx%d (image too small for D and block size)", rw, rh);
    const float lambda = c->prm.wls_lambda, sigma = c->prm.wls_sigma;
    if (!(lambda >= 0.0f) || !(sigma > 0.0f)) return fail(c, SV_ERR_INVALID_ARG, "need wls_lambda >= 0 and wls_sigma > 0");
    const int K = (vis && out8) ? 2 : 1, nimg = nf * K;
    const size_t nroi = (size_t)rw * rh;
    int rc;
    if ((rc = ensure(c, c->wls_lut, sizeof(float) * svk::WLS_LUT_SIZE))) return rc;
    if ((rc = ensure(c, c->wls_wh, sizeof(float) * nroi * nf))) return rc;
    if ((rc = ensure(c, c->wls_wv, sizeof(float) * nroi * nf))) return rc;
    if ((rc = ensure(c, c->wls_cur, sizeof(float) * nroi * nimg))) return rc;
    if ((rc = ensure(c, c->wls_tmp, sizeof(float) * nroi * nimg))) return rc;
    if (c->wls_lut_sigma != sigma) {
        // weight table of the fast global smoother: -exp(-sqrt(squared colour distance) / sigma), built
        // on the host in float (as the CPU statement does) so that both sides share every bit of it
        std::vector<float> lut(svk::WLS_LUT_SIZE);
        for (int i = 0; i < svk::WLS_LUT_SIZE; i++) lut[i] = -expf(-sqrtf((float)i) / sigma);
        HIPCHK(c, hipMemcpyAsync(c->wls_lut.p, lut.data(), sizeof(float) * lut.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // lut is a local
        c->wls_lut_sigma = sigma;
    }
    float* cur = (float*)c->wls_cur.p;
    float* tmp = (float*)c->wls_tmp.p;
    float* wh = (float*)c->wls_wh.p;
    float* wv = (float*)c->wls_wv.p;
    const dim3 b(256);
    hipLaunchKernelGGL(svk::k_wls_weights, dim3((rw + 255) / 256, rh, nf), b, 0, c->stream, guide, (const float*)c->wls_lut.p, wh,
                       wv, W, H, rx, ry, rw, rh);
    hipLaunchKernelGGL(svk::k_wls_load, dim3((rw + 255) / 256, rh, nimg), b, 0, c->stream, d16, vis, cur, W, H, rx, ry, rw, rh, K);
    float lam = lambda;
    for (int it = 0; it < 3; it++) {  // fastGlobalSmootherFilter defaults: 3 iterations, attenuation 0.25
        // one wave per 64 rows: few waves (a single frame) -> the lane-per-row form, whose latency is the solve chain alone;
        // many (a batch) -> the tile form, whose accesses are coalesced
        const dim3 hgrid((rh + svk::FGS_T - 1) / svk::FGS_T, nimg);
        if ((long)hgrid.x * hgrid.y <= 4L * c->n_cu)
            hipLaunchKernelGGL(svk::k_fgs_hpass, hgrid, dim3(64), 0, c->stream, cur, wh, tmp, rw, rh, K, lam);
        else
            hipLaunchKernelGGL(svk::k_fgs_hpass_tiles<32>, hgrid, dim3(64), 0, c->stream, cur, wh, tmp, rw, rh, K, lam);
        hipLaunchKernelGGL(svk::k_fgs_vpass, dim3((rw + 63) / 64, nimg), dim3(64), 0, c->stream, cur, wv, tmp, rw, rh, K, lam);
        lam *= 0.25f;
    }
    hipLaunchKernelGGL(svk::k_wls_store, dim3((W + 255) / 256, H, nimg), b, 0, c->stream, cur, out16, out8, W, H, rx, ry, rw, rh, K);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

// The matcher proper: device gray pair -> device disparity16.
// left / right: gray [nb][H][W], or with SV_FLAG_COST_RGB interleaved RGB [nb][H][W][3] (rgb = true;
// the two must agree: the gray entry points reject the flag, the RGB ones convert to gray without it)
int run_sgm(sv_ctx* c, const uint8_t* left, const uint8_t* right, int16_t* disp16, int nb = 1, bool rgb = false) {
    const sv_params& p = c->prm;
    const int W = c->W, H = c->H, D = p.num_disparities;
    const int cn = (p.flags & SV_FLAG_COST_RGB) ? 3 : 1;
    if ((cn == 3) != rgb)
        return fail(c, SV_ERR_INVALID_ARG, "SV_FLAG_COST_RGB needs the RGB entry points (sv_compute_rgb_*), and only those accept it");
    {   // never launch on a missing or short volume (a failed sv_set_batch / sv_set_params leaves core_ok = false)
        const CoreSizes z = core_sizes(c, nb);
        if (!c->core_ok || nb > c->batch || !c->stat_l.p || !c->stat_r.p || !c->C.p || !c->S.p || !c->rec.p || !c->disp_lr.p || !c->out.p ||
            c->stat_l.n < z.stat_l || c->stat_r.n < z.stat_r || c->C.n < z.vol || c->S.n < z.vol || c->rec.n < z.rec || c->disp_lr.n < z.plane)
            return fail(c, SV_ERR_NOMEM, "the context holds no volumes for %d frames (an earlier sv_set_batch / sv_set_params failed): call sv_set_batch again", nb);
    }
    const bool tm = c->timing && c->ev_ok;
    int e = 0;
    hipEvent_t* evs = c->ev[c->ev_calls % sv_ctx::EV_RING];
    auto mark = [&]() { if (tm) (void)hipEventRecord(evs[e++], c->stream); };
    int n_agg_ev = 0;
    hipEvent_t* evs_agg = c->ev_agg[c->ev_calls % sv_ctx::EV_RING];
    auto agg_mark = [&]() { if (tm && n_agg_ev < 8) (void)hipEventRecord(evs_agg[n_agg_ev++], c->stream); };
    // Fusing WTA into the last path saves one volume write + read but triples that launch's
    // instruction count per step; with few frames in flight the path kernels are latency-bound
    // (one wave per SIMD or less), so the fusion only pays once the batch fills the chip.
    // Measured crossover at 640x480 D=128: 6 frames x 512 chains; scaled by the chain count so that
    // one 1920x1080 frame (1664 chains per launch) counts as what it is.
    const int W1 = geo_w1(W, p), col0 = geo_col0(W, p), colN = geo_colN(W, p);
    const bool fuse = !c->debug && (c->force_fuse || (long)nb * W1 >= (long)c->fuse_wta_min_batch * 512);
    mark();
    // 1. prefilter + BT statistics
    {
        // four pixels per thread, one halo group on each side of a block: as few threads as cover a row, at most 256
        const int groups = (W + 3) / 4;
        const int T = groups + 2 <= 256 ? ((groups + 2 + 63) / 64) * 64 : 256;
        const int span = 4 * (T - 2);
        hipLaunchKernelGGL(svk::k_stats_planes, dim3((W + span - 1) / span, (H + svk::STATS_ROWS - 1) / svk::STATS_ROWS, 2 * nb * cn), dim3(T), 0, c->stream, left, right,
                           (uint2*)c->stat_l.p, (uint16_t*)c->stat_r.p, W, H, svk::cost_plane_pitch(W), p.pre_filter_cap, cn);
    }
    mark();
    // 2. matching cost volume C
    {
        int rc = launch_cost2(c, nb);
        if (rc) return rc;
    }
    mark();
    // 3. path aggregation.  Large batches: fused sweeps (four paths per pass over C, sv_sweep.hip.h).
    // Otherwise one launch per path: the first one initialises S and, when `fuse`, the last one
    // carries WTA and never stores S.
    bool wta_done = false;
    c->last_agg_launches = c->last_agg_sweeps = 0;
    {
        // The fused sweeps pipeline the bands of a frame through ceil(H / rows-per-band) workgroups and
        // pay the pipeline fill once per launch: they win once every pipeline has a frame to stream
        // (measured at 640x480 D=128 on 256 CUs = 32 pipelines: 24 frames 3000 vs 3060 frames/s for the
        // per-path kernels, 32 frames 3810 vs 3100), below that the per-path kernels are faster.
        // (with the 8-wave form chosen for small launches: from 5/8 of its pipelines, 18 frames there: 16 frames 2960 vs
        // 3090 frames/s for the per-path kernels, 20 frames 3540 vs 3020; scratch/perf_thresh.py)
        const int sw_pipes = sweep_pipes(c, sweep_cw_for(c, nb));
        // (round 3, after the sweeps lost a fifth of their instructions: from half of the 8-wave pipelines, 14 frames: 3630 vs 3460
        // frames/s, 16 frames 3810 vs 3530; 12 frames 3120 vs 3410; scratch/perf_thresh2.py)
        const int sw_auto = (D > 128 || c->sweep_cw) ? sw_pipes : (sweep_pipes(c, 7) + 1) / 2;
        const int sw_min = c->sweep_min_batch > 0 ? c->sweep_min_batch : sw_auto;
        // The sweeps take min(Lp(d), Lp(d-1) + P1, Lp(d+1) + P1) with the packed binary16 three-input minimum (sv_sweep.hip.h,
        // pk_min3), which orders u16 values below 0x7C00 like integers: every L (<= the largest cost) plus P1 has to stay
        // there.  sv_check_params allows costs up to 32767: the parameter sets in between take the per-path kernels.
        const long cmax = (long)((p.flags & SV_FLAG_COST_RGB) ? 3 : 1) * (2 * p.pre_filter_cap + 63) * p.block_size * p.block_size + p.P2;
        const bool sweep_f16_ok = cmax + p.P1 <= 0x7BFF;
        const bool sweep_ok = !c->debug && c->sweep_mode >= 0 && (c->sweep_mode > 0 || nb >= sw_min) && sweep_f16_ok;
        const bool sw1 = sweep_ok && (p.path_mask & 0x0Fu) == 0x0Fu;
        const bool sw2 = sw1 && (p.path_mask & 0xF0u) == 0xF0u;
        uint32_t rest = p.path_mask;
        bool first = true;
        // Below the sweeps' batch: every path of a chunk of frames in one launch, chunk after chunk.  A chunk holds as many
        // frames as give a path about two waves per CU (4 frames at 640x480 on 256 CUs: 3500 frames/s, against 3100-3250
        // for one launch per path over 8-17 frames; scratch/perf_batches.py), SV_FAN_CHUNK overrides.
        const long waves_per_frame = ((W1 > H ? W1 : H) + 3) / 4;
        int fan_chunk = c->fan_chunk > 0 ? c->fan_chunk : (int)((2L * c->n_cu + waves_per_frame / 2) / waves_per_frame);
        if (fan_chunk < 1) fan_chunk = 1;
        const bool fan_ok = !c->debug && !c->force_fuse && c->fanout_mode >= 0 && __builtin_popcount(p.path_mask) >= 2;
        // frames 0 .. n_sw - 1 through the sweeps, the rest (a few frames beyond whole rounds of the pipelines) the other way
        const int n_sw = (sw2 && fan_ok && c->sweep_mode == 0 && c->sweep_min_batch == 0) ? sweep_share(c, nb) : nb;
        if (sw1) {
            int rc = launch_sweep(c, false, n_sw);
            if (rc) return rc;
            rest &= ~0x0Fu;
            first = false;
            c->last_agg_launches++; c->last_agg_sweeps++;
            agg_mark();
        }
        if (sw2) {
            int rc = launch_sweep(c, true, n_sw);
            if (rc) return rc;
            rest = 0;
            wta_done = true;
            c->last_agg_launches++; c->last_agg_sweeps++;
            agg_mark();
            if (n_sw < nb) {
                rc = run_paths_fanout(c, n_sw, nb - n_sw, fan_chunk);
                if (rc) return rc;
                agg_mark();
            }
        }
        const int n_rest = __builtin_popcount(rest);
        const bool fan = fan_ok && first && n_rest >= 2;
        if (fan) {
            int rc = run_paths_fanout(c, 0, nb, fan_chunk);
            if (rc) return rc;
            rest = 0;
            wta_done = true;
            agg_mark();
        }
        if (rest) {
            int last = 7;
            while (!(rest & (1u << last))) last--;
            for (int dir = 0; dir < 8; dir++)
                if (rest & (1u << dir)) {
                    const bool fl = dir == last && fuse;
                    launch_agg(c, dir, (first ? 1 : 0) | (fl ? 2 : 0), nb);
                    c->last_agg_launches++;
                    agg_mark();
                    first = false;
                    wta_done = wta_done || fl;
                }
        }
    }
    mark();
    // 4. WTA as its own kernel when it was not fused into the last path / second sweep
    if (!wta_done) launch_wta(c, nb);
    mark();
    // 5. LR check, median, speckle
    {
        // without the LR check the same kernel just unpacks the records (threshold that never trips)
        int thr = p.disp12_max_diff;
        if (p.flags & SV_FLAG_LR_OPENCV) thr = thr > 0 ? thr : 1;
        if (p.flags & SV_FLAG_NO_LRCHECK) thr = 1 << 20;
        // production path with the median on: one kernel for both, the checked map stays in LDS (a row of keys
        // and three checked rows: 10 bytes per column).  The debug variants keep the two kernels (and disp_lr).
        const size_t lds_fused = (size_t)W * 10;
        // (a workgroup walks its 16 + 2 rows one after the other: only with two workgroups per CU or more in the
        // launch; a single 640x480 frame is faster through the 480 + 1200 small workgroups of the two kernels)
        const long tail_wgs = (long)((H + svk::LRM_ROWS - 1) / svk::LRM_ROWS) * nb;
        const bool fused_tail = !(p.flags & SV_FLAG_NO_MEDIAN) && (c->debug_mode == 0 || c->debug_mode == 3) && lds_fused <= 64 * 1024 &&
                                tail_wgs >= 2L * c->n_cu;
        if (fused_tail) {
            hipLaunchKernelGGL(svk::k_lr_median, dim3((H + svk::LRM_ROWS - 1) / svk::LRM_ROWS, nb), dim3(256), lds_fused, c->stream,
                               (const uint2*)c->rec.p, disp16, W, H, col0, colN, p.min_disparity, thr, (p.flags & SV_FLAG_MEDIAN_REPLICATE) ? 1 : 0);
        } else {
            if ((size_t)W * 4 > 48 * 1024 && !c->attr_lrcheck) {
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&svk::k_lrcheck), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                c->attr_lrcheck = true;
            }
            hipLaunchKernelGGL(svk::k_lrcheck, dim3(H, nb), dim3(256), (size_t)W * 4, c->stream, (const uint2*)c->rec.p,
                               (int16_t*)c->disp_lr.p, W, col0, colN, p.min_disparity, thr);
            const int16_t* cur = (const int16_t*)c->disp_lr.p;
            if (!(p.flags & SV_FLAG_NO_MEDIAN)) {
                hipLaunchKernelGGL(svk::k_median3, dim3((W + 255) / 256, H, nb), dim3(256), 0, c->stream, cur, disp16, W, H,
                                   (p.flags & SV_FLAG_MEDIAN_REPLICATE) ? 1 : 0);
            } else {
                HIPCHK(c, hipMemcpyAsync(disp16, cur, (size_t)W * H * 2 * nb, hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    if (p.speckle_window_size > 0) {
        int rc = speckle_dev(c, disp16, p.speckle_window_size, 16 * p.speckle_range, nb);
        if (rc) return rc;
    }
    mark();
    HIPCHK(c, hipGetLastError());
    if (tm) {
        c->ev_agg_n[c->ev_calls % sv_ctx::EV_RING] = n_agg_ev;
        c->ev_calls++;
    }
    return SV_OK;
}

}  // namespace

extern "C" {

const char* sv_version(void) { return "libsv_hip 0.1 (gfx950)"; }

const char* sv_last_error(const sv_ctx* ctx) { return ctx ? ctx->err : k_noctx; }

int sv_default_params(sv_params* p, int num_disparities, int paths) {
    if (!p || (paths != 5 && paths != 8)) return SV_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->min_disparity = 0;
    p->num_disparities = num_disparities;
    p->block_size = 7;
    p->P1 = 8 * 49;
    p->P2 = 32 * 49;
    p->pre_filter_cap = 63;
    p->uniqueness_ratio = 15;
    p->disp12_max_diff = 0;
    p->speckle_window_size = 0;
    p->speckle_range = 0;
    p->path_mask = paths == 8 ? SV_PATHS_8 : SV_PATHS_5;
    p->flags = paths == 8 ? SV_FLAGS_HH : SV_FLAGS_REPO;
    p->wls_lambda = 4500.0f;
    p->wls_sigma = 1.0f;
    return SV_OK;
}

int sv_create(sv_ctx** out, int device, int width, int height, const sv_params* params) {
    if (!out) return SV_ERR_INVALID_ARG;
    *out = nullptr;
    sv_ctx* c = new (std::nothrow) sv_ctx();
    if (!c) return SV_ERR_NOMEM;
    *out = c;  // returned even on failure so the caller can read sv_last_error, then sv_destroy
    c->device = device;
    c->W = width;
    c->H = height;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, SV_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(c, SV_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
    int rc = check_params(c, params, width, height);
    if (rc) return rc;
    c->prm = *params;
    DevGuard g(device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, SV_ERR_NO_DEVICE, "device %d is %s; libsv_hip is built for gfx950 only", device, prop.gcnArchName);
    for (int r = 0; r < sv_ctx::EV_RING; r++)
        for (int i = 0; i < sv_ctx::EV_PER; i++) HIPCHK(c, hipEventCreate(&c->ev[r][i]));
    for (int r = 0; r < sv_ctx::EV_RING; r++)
        for (int i = 0; i < 8; i++) HIPCHK(c, hipEventCreate(&c->ev_agg[r][i]));
    c->ev_ok = true;
    if (const char* e = getenv("SV_FUSE_WTA_MIN_BATCH")) c->fuse_wta_min_batch = atoi(e);  // tuning knobs
    if (const char* e = getenv("SV_SWEEP_MIN_BATCH")) c->sweep_min_batch = atoi(e);
    if (const char* e = getenv("SV_SWEEP_MODE")) c->sweep_mode = atoi(e);
    if (const char* e = getenv("SV_FAN_CHUNK")) c->fan_chunk = atoi(e);
    if (const char* e = getenv("SV_SWEEP_FAULT")) c->sweep_fault = atoi(e);   // tests: the next n sweep launches report a timeout
    if (const char* e = getenv("SV_SWEEP_CW")) c->sweep_cw = atoi(e) == 7 ? 7 : (atoi(e) == 15 ? 15 : 0);
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("SV_SWEEP_NCU")) c->n_cu = atoi(e) > 0 ? atoi(e) : c->n_cu;  // tests: few groups, many frames each
    HIPCHK(c, hipHostMalloc((void**)&c->sw_error, sizeof(unsigned int), hipHostMallocMapped));
    *c->sw_error = 0;
    HIPCHK(c, static_scratch_lookup(device, &c->static_scratch));  // this device's copy of the static scratch (sv_filter_speckles_static_dev)
    return alloc_core(c, 1, c->prm);
}

int sv_destroy(sv_ctx* c) {
    if (!c) return SV_ERR_INVALID_ARG;
    {
        DevGuard g(c->device);
        if (g.ok) {
            (void)hipStreamSynchronize(c->stream);
            Buf* all[] = {&c->img_l, &c->img_r, &c->stat_l, &c->stat_r, &c->C, &c->S, &c->rec,
                          &c->disp_lr, &c->out, &c->rgb_l, &c->rgb_r, &c->rect_l, &c->rect_r, &c->gray_l, &c->gray_r,
                          &c->maps[0], &c->maps[1], &c->maps[2], &c->maps[3], &c->mapper, &c->mapper_lr[0], &c->mapper_lr[1], &c->depth, &c->d16tmp, &c->vis,
                          &c->lab, &c->cnt, &c->wls_lut, &c->wls_wh, &c->wls_wv, &c->wls_cur, &c->wls_tmp, &c->wls_g,
                          &c->wls_v, &c->wls_o16, &c->wls_o8, &c->world_cart, &c->world_cyl, &c->blf_a, &c->blf_b, &c->blf_mm, &c->blf_o16, &c->blf_o8, &c->blf_tab, &c->blf_tmp, &c->blf_gray, &c->sw_handoff, &c->sw_progress, &c->sw_sink, &c->sw_prof, &c->cost_sink};
            for (Buf* b : all) release(*b);
            if (c->sw_error) (void)hipHostFree(c->sw_error);
            if (c->slots_ok)
                for (auto& sl : c->slot) {
                    if (sl.in) (void)hipHostFree(sl.in);
                    if (sl.out) (void)hipHostFree(sl.out);
                    if (sl.done) (void)hipEventDestroy(sl.done);
                }
            for (int i = 0; i < 8; i++) release(c->Lbuf[i]);
            if (c->bslots_ok) {
                if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
                if (c->down_stream) { (void)hipStreamSynchronize(c->down_stream); (void)hipStreamDestroy(c->down_stream); }
                for (auto& b : c->bslot) {
                    release(b.in_l); release(b.in_r); release(b.out);
                    if (b.up) (void)hipEventDestroy(b.up);
                    if (b.computed) (void)hipEventDestroy(b.computed);
                    if (b.done) (void)hipEventDestroy(b.done);
                }
            }

            if (c->ev_ok)
                for (int r = 0; r < sv_ctx::EV_RING; r++)
                    for (int i = 0; i < sv_ctx::EV_PER; i++)
                        if (c->ev[r][i]) (void)hipEventDestroy(c->ev[r][i]);
            for (int r = 0; r < sv_ctx::EV_RING; r++)
                for (int i = 0; i < 8; i++)
                    if (c->ev_agg[r][i]) (void)hipEventDestroy(c->ev_agg[r][i]);
        }
    }
    delete c;
    return SV_OK;
}

int sv_set_params(sv_ctx* c, const sv_params* params) {
    if (!c) return SV_ERR_INVALID_ARG;
    int rc = check_params(c, params, c->W, c->H);
    if (rc) return rc;
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const bool regeom = params->num_disparities != c->prm.num_disparities || params->min_disparity != c->prm.min_disparity ||
                        ((params->flags ^ c->prm.flags) & SV_FLAG_COST_RGB);  // statistics per colour plane
    if (regeom) HIPCHK(c, hipStreamSynchronize(c->stream));
    const sv_params old = c->prm;
    c->prm = *params;
    return regeom ? alloc_core(c, c->batch, old) : SV_OK;
}

int sv_get_params(const sv_ctx* c, sv_params* params) {
    if (!c || !params) return SV_ERR_INVALID_ARG;
    *params = c->prm;
    return SV_OK;
}

int sv_set_stream(sv_ctx* c, void* hip_stream) {
    if (!c) return SV_ERR_INVALID_ARG;
    c->stream = (hipStream_t)hip_stream;
    return SV_OK;
}

int sv_compute_gray_dev(sv_ctx* c, const uint8_t* left_dev, const uint8_t* right_dev, int16_t* disp16_dev) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left_dev || !right_dev || !disp16_dev) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    return run_sgm(c, left_dev, right_dev, disp16_dev);
}

int sv_set_batch(sv_ctx* c, int max_frames) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (max_frames < 1 || max_frames > 2048) return fail(c, SV_ERR_INVALID_ARG, "batch must be in [1,2048] (got %d)", max_frames);
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (max_frames != c->batch) HIPCHK(c, hipStreamSynchronize(c->stream));
    const int old = c->batch;
    c->batch = max_frames;
    return alloc_core(c, old, c->prm);
}

int sv_compute_gray_batch_dev(sv_ctx* c, int n_frames, const uint8_t* left_dev, const uint8_t* right_dev,
                              int16_t* disp16_dev) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left_dev || !right_dev || !disp16_dev) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    if (n_frames < 1 || n_frames > c->batch)
        return fail(c, SV_ERR_INVALID_ARG, "n_frames %d outside [1, batch=%d] (sv_set_batch)", n_frames, c->batch);
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    return run_sgm(c, left_dev, right_dev, disp16_dev, n_frames);
}

int sv_compute_rgb_batch_dev(sv_ctx* c, int n_frames, const uint8_t* left, const uint8_t* right, int rectify, uint8_t* rect_left,
                             int16_t* disp16) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left || !right || !disp16) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    if (n_frames < 1 || n_frames > c->batch)
        return fail(c, SV_ERR_INVALID_ARG, "n_frames %d outside [1, batch=%d] (sv_set_batch)", n_frames, c->batch);
    if (rectify && !c->have_maps) return fail(c, SV_ERR_NO_MAPS, "rectification maps not uploaded (sv_upload_maps_host / sv_init_rectify_map)");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const size_t W = c->W, H = c->H, n = n_frames;
    int rc;
    const uint8_t *pl = left, *pr = right;
    if (rectify) {
        // the two cv::remap calls of rectifyImages (stereoCamera.cpp:573-574), every frame of the batch in one launch
        if (!rect_left) {
            if ((rc = ensure(c, c->rect_l, n * W * H * 3))) return rc;
            rect_left = (uint8_t*)c->rect_l.p;
        }
        if ((rc = ensure(c, c->rect_r, n * W * H * 3))) return rc;
        const dim3 grid((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)n);
        hipLaunchKernelGGL(svk::k_remap_rgb, grid, dim3(256), 0, c->stream, left, rect_left, (const float*)c->maps[0].p,
                           (const float*)c->maps[1].p, c->W, c->H);
        hipLaunchKernelGGL(svk::k_remap_rgb, grid, dim3(256), 0, c->stream, right, (uint8_t*)c->rect_r.p, (const float*)c->maps[2].p,
                           (const float*)c->maps[3].p, c->W, c->H);
        HIPCHK(c, hipGetLastError());
        pl = rect_left;
        pr = (const uint8_t*)c->rect_r.p;
    } else if (rect_left) {
        HIPCHK(c, hipMemcpyAsync(rect_left, left, n * W * H * 3, hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->prm.flags & SV_FLAG_COST_RGB) return run_sgm(c, pl, pr, disp16, n_frames, true);
    if ((rc = ensure(c, c->gray_l, n * W * H))) return rc;
    if ((rc = ensure(c, c->gray_r, n * W * H))) return rc;
    const size_t npx = n * W * H;
    hipLaunchKernelGGL(svk::k_gray, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, pl, (uint8_t*)c->gray_l.p, npx);
    hipLaunchKernelGGL(svk::k_gray, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, pr, (uint8_t*)c->gray_r.p, npx);
    HIPCHK(c, hipGetLastError());
    return run_sgm(c, (const uint8_t*)c->gray_l.p, (const uint8_t*)c->gray_r.p, disp16, n_frames);
}

int sv_compute_gray_host(sv_ctx* c, const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int16_t* disp16, size_t ds) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left || !right || !disp16) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    const size_t W = c->W, H = c->H;
    if (ls < W || rs < W || ds < 2 * W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int rc;
    if ((rc = ensure(c, c->img_l, W * H))) return rc;
    if ((rc = ensure(c, c->img_r, W * H))) return rc;
    HIPCHK(c, copy2d(c->img_l.p, W, left, ls, W, H, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, copy2d(c->img_r.p, W, right, rs, W, H, hipMemcpyHostToDevice, c->stream));
    if ((rc = run_sgm(c, (const uint8_t*)c->img_l.p, (const uint8_t*)c->img_r.p, (int16_t*)c->out.p))) return rc;
    HIPCHK(c, copy2d(disp16, ds, c->out.p, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_sweep_error(c);
}

int sv_submit_gray_host(sv_ctx* c, const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int* ticket) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left || !right || !ticket) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    const size_t W = c->W, H = c->H;
    if (ls < W || rs < W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (!c->slots_ok) {
        for (auto& sl : c->slot) {
            HIPCHK(c, hipHostMalloc((void**)&sl.in, 2 * W * H, hipHostMallocDefault));
            HIPCHK(c, hipHostMalloc((void**)&sl.out, 2 * W * H, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
        c->slots_ok = true;
    }
    sv_ctx::Slot& sl = c->slot[c->next_ticket % sv_ctx::N_SLOTS];
    if (sl.busy) return fail(c, SV_ERR_INVALID_ARG, "%d pairs already in flight: sv_wait_host the oldest ticket first", sv_ctx::N_SLOTS);
    int rc;
    if ((rc = ensure(c, c->img_l, W * H))) return rc;
    if ((rc = ensure(c, c->img_r, W * H))) return rc;
    // into pinned memory: the copies below are then truly asynchronous
    if (ls == W && rs == W) {
        memcpy(sl.in, left, W * H);
        memcpy(sl.in + W * H, right, W * H);
    } else {
        for (size_t y = 0; y < H; y++) {
            memcpy(sl.in + y * W, left + y * ls, W);
            memcpy(sl.in + (H + y) * W, right + y * rs, W);
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->img_l.p, sl.in, W * H, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->img_r.p, sl.in + W * H, W * H, hipMemcpyHostToDevice, c->stream));
    if ((rc = run_sgm(c, (const uint8_t*)c->img_l.p, (const uint8_t*)c->img_r.p, (int16_t*)c->out.p))) return rc;
    HIPCHK(c, hipMemcpyAsync(sl.out, c->out.p, 2 * W * H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(sl.done, c->stream));
    sl.busy = true;
    *ticket = (int)(c->next_ticket++ & 0x7FFFFFFF);
    return SV_OK;
}

int sv_wait_host(sv_ctx* c, int ticket, int16_t* disp16, size_t ds) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!disp16) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    const size_t W = c->W, H = c->H;
    if (ds < 2 * W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    if (!c->slots_ok || ticket < 0) return fail(c, SV_ERR_INVALID_ARG, "unknown ticket %d", ticket);
    sv_ctx::Slot& sl = c->slot[ticket % sv_ctx::N_SLOTS];
    const long newest = c->next_ticket - 1;
    if (!sl.busy || (long)ticket > (newest & 0x7FFFFFFF) || (newest & 0x7FFFFFFF) - ticket >= sv_ctx::N_SLOTS)
        return fail(c, SV_ERR_INVALID_ARG, "ticket %d is not in flight", ticket);
    DevGuard g(c->device);
    HIPCHK(c, hipEventSynchronize(sl.done));
    if (ds == 2 * W) memcpy(disp16, sl.out, 2 * W * H);
    else
        for (size_t y = 0; y < H; y++) memcpy((char*)disp16 + y * ds, sl.out + y * W, 2 * W);
    sl.busy = false;
    return check_sweep_error(c);
}

int sv_submit_gray_batch_host(sv_ctx* c, int n_frames, const uint8_t* const* left, size_t ls, const uint8_t* const* right, size_t rs,
                              int16_t* const* disp16, size_t ds, int* ticket) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left || !right || !disp16 || !ticket) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if (n_frames < 1 || n_frames > c->batch)
        return fail(c, SV_ERR_INVALID_ARG, "n_frames %d outside [1, batch=%d] (sv_set_batch)", n_frames, c->batch);
    const size_t W = c->W, H = c->H, n = n_frames;
    if (ls < W || rs < W || ds < 2 * W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    for (size_t i = 0; i < n; i++)
        if (!left[i] || !right[i] || !disp16[i]) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer (frame %zu)", i);
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (!c->bslots_ok) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
        for (auto& b : c->bslot) {
            HIPCHK(c, hipEventCreateWithFlags(&b.up, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&b.computed, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
        }
        c->bslots_ok = true;
    }
    sv_ctx::BatchSlot& b = c->bslot[c->next_batch_ticket % SV_BATCH_SLOTS];
    if (b.busy) return fail(c, SV_ERR_INVALID_ARG, "%d batches already in flight: sv_wait_batch_host the oldest ticket first", SV_BATCH_SLOTS);
    int rc;
    // the slot's device buffers were last used by a batch that has been waited for
    if ((rc = ensure(c, b.in_l, n * W * H))) return rc;
    if ((rc = ensure(c, b.in_r, n * W * H))) return rc;
    if ((rc = ensure(c, b.out, n * W * H * 2))) return rc;
    // Uploads.  Frames that follow each other densely in host memory (one array of frames) travel as one copy.
    auto upload = [&](const uint8_t* const* img, size_t stride, uint8_t* dst) -> hipError_t {
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            if (stride == W)
                while (j < n && img[j] == img[j - 1] + W * H) j++;
            hipError_t e = copy2d(dst + i * W * H, W, img[i], stride, W, H * (stride == W ? j - i : 1), hipMemcpyHostToDevice, c->up_stream);
            if (e != hipSuccess) return e;
            i = stride == W ? j : i + 1;
        }
        return hipSuccess;
    };
    HIPCHK(c, upload(left, ls, (uint8_t*)b.in_l.p));
    HIPCHK(c, upload(right, rs, (uint8_t*)b.in_r.p));
    HIPCHK(c, hipEventRecord(b.up, c->up_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, b.up, 0));
    if ((rc = run_sgm(c, (const uint8_t*)b.in_l.p, (const uint8_t*)b.in_r.p, (int16_t*)b.out.p, n_frames))) return rc;
    HIPCHK(c, hipEventRecord(b.computed, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->down_stream, b.computed, 0));
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        if (ds == 2 * W)
            while (j < n && disp16[j] == disp16[j - 1] + W * H) j++;
        HIPCHK(c, copy2d(disp16[i], ds, (const int16_t*)b.out.p + i * W * H, 2 * W, 2 * W, H * (ds == 2 * W ? j - i : 1), hipMemcpyDeviceToHost,
                         c->down_stream));
        i = ds == 2 * W ? j : i + 1;
    }
    HIPCHK(c, hipEventRecord(b.done, c->down_stream));
    b.busy = true;
    b.dst.assign(disp16, disp16 + n);
    b.ds = ds;
    *ticket = (int)(c->next_batch_ticket++ & 0x7FFFFFFF);
    return SV_OK;
}

int sv_wait_batch_host(sv_ctx* c, int ticket) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!c->bslots_ok || ticket < 0) return fail(c, SV_ERR_INVALID_ARG, "unknown batch ticket %d", ticket);
    sv_ctx::BatchSlot& b = c->bslot[ticket % SV_BATCH_SLOTS];
    const long newest = (c->next_batch_ticket - 1) & 0x7FFFFFFF;
    if (!b.busy || (long)ticket > newest || newest - ticket >= SV_BATCH_SLOTS) return fail(c, SV_ERR_INVALID_ARG, "batch ticket %d is not in flight", ticket);
    DevGuard g(c->device);
    HIPCHK(c, hipEventSynchronize(b.done));
    b.busy = false;
    if (check_sweep_error(c) != SV_OK) {
        // The batch's inputs are still in the slot's device buffers: run it again, now on the per-path kernels
        // (check_sweep_error switched the context over), and bring the results down on the compute stream.
        const size_t W = c->W, H = c->H, n = b.dst.size();
        int rc = run_sgm(c, (const uint8_t*)b.in_l.p, (const uint8_t*)b.in_r.p, (int16_t*)b.out.p, (int)n);
        if (rc) return rc;
        for (size_t i = 0; i < n; i++)
            HIPCHK(c, copy2d(b.dst[i], b.ds, (const int16_t*)b.out.p + i * W * H, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return check_sweep_error(c);
    }
    return SV_OK;
}

int sv_host_alloc(void** ptr, size_t bytes) {
    if (!ptr || bytes == 0) return SV_ERR_INVALID_ARG;
    return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? SV_OK : SV_ERR_NOMEM;
}
int sv_host_free(void* ptr) { return hipHostFree(ptr) == hipSuccess ? SV_OK : SV_ERR_HIP; }

int sv_gray_from_rgb_dev(sv_ctx* c, const uint8_t* rgb_dev, uint8_t* gray_dev) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!rgb_dev || !gray_dev) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    DevGuard g(c->device);
    const size_t n = (size_t)c->W * c->H;
    hipLaunchKernelGGL(svk::k_gray, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rgb_dev, gray_dev, n);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

int sv_upload_maps_host(sv_ctx* c, const float* mlx, const float* mly, const float* mrx, const float* mry) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!mlx || !mly || !mrx || !mry) return fail(c, SV_ERR_INVALID_ARG, "NULL map pointer");
    DevGuard g(c->device);
    const size_t bytes = (size_t)c->W * c->H * 4;
    const float* src[4] = {mlx, mly, mrx, mry};
    for (int i = 0; i < 4; i++) {
        int rc = ensure(c, c->maps[i], bytes);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->maps[i].p, src[i], bytes, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_maps = c->have_map[0] = c->have_map[1] = true;
    return SV_OK;
}

int sv_init_rectify_map(sv_ctx* c, int which, const double* K, const double* dist, int n_dist, const double* R, const double* P) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (which < 0 || which > 1) return fail(c, SV_ERR_INVALID_ARG, "which must be 0 (left) or 1 (right)");
    svk::CamModel cm;
    int rc = cam_model(c, K, dist, n_dist, R, P, true, &cm);
    if (rc) return rc;
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const size_t bytes = (size_t)c->W * c->H * 4;
    if ((rc = ensure(c, c->maps[2 * which], bytes))) return rc;
    if ((rc = ensure(c, c->maps[2 * which + 1], bytes))) return rc;
    const size_t lds = (size_t)c->W * 3 * sizeof(double);
    if (lds > 160 * 1024) return fail(c, SV_ERR_UNSUPPORTED, "image width %d too large for the row buffer of the map kernel", c->W);
    if (lds > 48 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&svk::k_init_rectify_map), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(svk::k_init_rectify_map, dim3(c->H), dim3(256), lds, c->stream,
                       (float*)c->maps[2 * which].p, (float*)c->maps[2 * which + 1].p, c->W, cm);
    HIPCHK(c, hipGetLastError());
    c->have_map[which] = true;
    c->have_maps = c->have_map[0] && c->have_map[1];
    return SV_OK;
}

int sv_update_mapper(sv_ctx* c, int which, const double* K, const double* dist, int n_dist, const double* R, const double* P) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (which < 0 || which > 1) return fail(c, SV_ERR_INVALID_ARG, "which must be 0 (left) or 1 (right)");
    svk::CamModel cm;
    int rc = cam_model(c, K, dist, n_dist, R, P, false, &cm);
    if (rc) return rc;
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if ((rc = ensure(c, c->mapper_lr[which], (size_t)c->W * c->H * 8))) return rc;
    hipLaunchKernelGGL(svk::k_undistort_grid, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream,
                       (float2*)c->mapper_lr[which].p, c->W, c->H, cm);
    HIPCHK(c, hipGetLastError());
    c->have_mapper[which] = true;
    return SV_OK;
}

int sv_get_maps_host(sv_ctx* c, int which, float* mapx, float* mapy) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (which < 0 || which > 1 || !mapx || !mapy) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    if (!c->have_map[which]) return fail(c, SV_ERR_NO_MAPS, "no rectification map for this camera yet");
    DevGuard g(c->device);
    const size_t bytes = (size_t)c->W * c->H * 4;
    HIPCHK(c, hipMemcpyAsync(mapx, c->maps[2 * which].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(mapy, c->maps[2 * which + 1].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_get_mapper_host(sv_ctx* c, int which, float* mapper) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (which < 0 || which > 1 || !mapper) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    if (!c->have_mapper[which]) return fail(c, SV_ERR_NO_MAPS, "no mapper for this camera yet (sv_update_mapper)");
    DevGuard g(c->device);
    HIPCHK(c, hipMemcpyAsync(mapper, c->mapper_lr[which].p, (size_t)c->W * c->H * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_remap_rgb_dev(sv_ctx* c, int which, const uint8_t* src, uint8_t* dst) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!src || !dst || which < 0 || which > 1) return fail(c, SV_ERR_INVALID_ARG, "bad remap argument");
    if (!c->have_maps) return fail(c, SV_ERR_NO_MAPS, "rectification maps not uploaded (sv_upload_maps_host)");
    DevGuard g(c->device);
    hipLaunchKernelGGL(svk::k_remap_rgb, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream, src, dst,
                       (const float*)c->maps[2 * which].p, (const float*)c->maps[2 * which + 1].p, c->W, c->H);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

int sv_compute_rgb_vis_host(sv_ctx* c, const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int rectify,
                            int16_t* disp16, size_t ds, uint8_t* vis, size_t vs, int vis_scale) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!left || !right || !disp16) return fail(c, SV_ERR_INVALID_ARG, "NULL image pointer");
    const size_t W = c->W, H = c->H;
    if (ls < 3 * W || rs < 3 * W || ds < 2 * W || (vis && vs < W)) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    if (rectify && !c->have_maps) return fail(c, SV_ERR_NO_MAPS, "rectification maps not uploaded (sv_upload_maps_host)");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int rc;
    Buf* need3[] = {&c->rgb_l, &c->rgb_r, &c->rect_l, &c->rect_r};
    for (Buf* b : need3)
        if ((rc = ensure(c, *b, W * H * 3))) return rc;
    if ((rc = ensure(c, c->gray_l, W * H))) return rc;
    if ((rc = ensure(c, c->gray_r, W * H))) return rc;
    if (vis && (rc = ensure(c, c->vis, W * H))) return rc;
    Buf& dl = rectify ? c->rgb_l : c->rect_l;
    Buf& dr = rectify ? c->rgb_r : c->rect_r;
    HIPCHK(c, copy2d(dl.p, 3 * W, left, ls, 3 * W, H, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, copy2d(dr.p, 3 * W, right, rs, 3 * W, H, hipMemcpyHostToDevice, c->stream));
    if (rectify) {
        if ((rc = sv_remap_rgb_dev(c, 0, (const uint8_t*)c->rgb_l.p, (uint8_t*)c->rect_l.p))) return rc;
        if ((rc = sv_remap_rgb_dev(c, 1, (const uint8_t*)c->rgb_r.p, (uint8_t*)c->rect_r.p))) return rc;
    }
    if (c->prm.flags & SV_FLAG_COST_RGB) {
        // matchSGBM semantics: the matcher runs on the colour pair itself (StereoMatcher.cpp:412)
        if ((rc = run_sgm(c, (const uint8_t*)c->rect_l.p, (const uint8_t*)c->rect_r.p, (int16_t*)c->out.p, 1, true))) return rc;
    } else {
        // matchSGBMCUDA semantics: cvtColor on both images first (StereoMatcher.cpp:429-430)
        if ((rc = sv_gray_from_rgb_dev(c, (const uint8_t*)c->rect_l.p, (uint8_t*)c->gray_l.p))) return rc;
        if ((rc = sv_gray_from_rgb_dev(c, (const uint8_t*)c->rect_r.p, (uint8_t*)c->gray_r.p))) return rc;
        if ((rc = run_sgm(c, (const uint8_t*)c->gray_l.p, (const uint8_t*)c->gray_r.p, (int16_t*)c->out.p))) return rc;
    }
    HIPCHK(c, copy2d(disp16, ds, c->out.p, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
    if (vis) {   // getDisparityVis(disp16, vis, scale) from the map that is still on the device
        if ((rc = sv_disparity_vis_dev(c, (const int16_t*)c->out.p, (uint8_t*)c->vis.p, vis_scale))) return rc;
        HIPCHK(c, copy2d(vis, vs, c->vis.p, W, W, H, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_sweep_error(c);
}

int sv_compute_rgb_host(sv_ctx* c, const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int rectify,
                        int16_t* disp16, size_t ds) {
    return sv_compute_rgb_vis_host(c, left, ls, right, rs, rectify, disp16, ds, nullptr, 0, 0);
}

int sv_get_rectified_host(sv_ctx* c, int which, uint8_t* rgb, size_t stride) {
    if (!c) return SV_ERR_INVALID_ARG;
    const size_t W = c->W, H = c->H;
    Buf& b = which ? c->rect_r : c->rect_l;
    if (!rgb || stride < 3 * W || which < 0 || which > 1) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    if (!b.p) return fail(c, SV_ERR_INVALID_ARG, "no rectified image yet (call sv_compute_rgb_host first)");
    DevGuard g(c->device);
    HIPCHK(c, copy2d(rgb, stride, b.p, 3 * W, 3 * W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_disparity_vis_dev(sv_ctx* c, const int16_t* d16, uint8_t* vis, int scale) {
    return sv_disparity_vis_batch_dev(c, 1, d16, vis, scale);
}

int sv_disparity_vis_batch_dev(sv_ctx* c, int n_frames, const int16_t* d16, uint8_t* vis, int scale) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !vis || n_frames < 1) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    DevGuard g(c->device);
    const size_t n = (size_t)c->W * c->H * n_frames;
    hipLaunchKernelGGL(svk::k_vis, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d16, vis, n, scale);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

int sv_disparity_vis_host(sv_ctx* c, const int16_t* d16, uint8_t* vis, int scale) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !vis) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    DevGuard g(c->device);
    const size_t n = (size_t)c->W * c->H;
    int rc;
    if ((rc = ensure(c, c->d16tmp, n * 2))) return rc;
    if ((rc = ensure(c, c->vis, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d16tmp.p, d16, n * 2, hipMemcpyHostToDevice, c->stream));
    if ((rc = sv_disparity_vis_dev(c, (const int16_t*)c->d16tmp.p, (uint8_t*)c->vis.p, scale))) return rc;
    HIPCHK(c, hipMemcpyAsync(vis, c->vis.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_filter_speckles_dev(sv_ctx* c, int16_t* d16, int max_size, int max_diff) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || max_size < 0 || max_diff < 0) return fail(c, SV_ERR_INVALID_ARG, "bad speckle argument");
    DevGuard g(c->device);
    return speckle_dev(c, d16, max_size, max_diff, 1);
}

int sv_filter_speckles_static_dev(sv_ctx* c, int16_t* d16, int max_size, int max_diff) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || max_size < 0 || max_diff < 0) return fail(c, SV_ERR_INVALID_ARG, "bad speckle argument");
    const size_t npx = (size_t)c->W * c->H;
    if (npx > (size_t)SV_STATIC_MAX_PIXELS)
        return fail(c, SV_ERR_UNSUPPORTED, "sv_filter_speckles_static_dev: %dx%d = %zu pixels exceed the static scratch of %d pixels; use sv_filter_speckles_dev",
                    c->W, c->H, npx, SV_STATIC_MAX_PIXELS);
    if (!c->static_scratch) return fail(c, SV_ERR_HIP, "no static scratch address for device %d", c->device);
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int* const lab = c->static_scratch;   // looked up in sv_create; never the symbol's name: that is not a device address in host code
    return speckle_launch(c, d16, lab, lab + SV_STATIC_MAX_PIXELS, c->W, c->H, max_size, max_diff, 1, c->stream);
}

int sv_depth_dev(sv_ctx* c, const int16_t* d16, const float* mapper, const double* Q, const double* R, float* depth) {
    return sv_depth_batch_dev(c, 1, d16, mapper, Q, R, depth);
}

int sv_depth_batch_dev(sv_ctx* c, int n_frames, const int16_t* d16, const float* mapper, const double* Q, const double* R,
                       float* depth) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !Q || !R || !depth || n_frames < 1 || n_frames > 65535) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    if (!mapper) {
        if (!c->have_mapper[0]) return fail(c, SV_ERR_NO_MAPS, "no mapper: pass one or call sv_update_mapper(ctx, 0, ...)");
        mapper = (const float*)c->mapper_lr[0].p;
    }
    DevGuard g(c->device);
    svk::DepthCoef k;
    k.r02 = (float)R[2]; k.r12 = (float)R[5]; k.r22 = (float)R[8];
    k.q00 = (float)Q[0]; k.q03 = (float)Q[3]; k.q11 = (float)Q[5]; k.q13 = (float)Q[7];
    k.q23 = (float)Q[11]; k.q32 = (float)Q[14]; k.q33 = (float)Q[15];
    hipLaunchKernelGGL(svk::k_depth, dim3((c->W + 255) / 256, c->H, n_frames), dim3(256), 0, c->stream, d16, (const float2*)mapper, depth,
                       c->W, c->H, k);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

int sv_depth_host(sv_ctx* c, const int16_t* d16, size_t dstride, const float* mapper, const double* Q, const double* R,
                  float* depth, size_t ostride) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !Q || !R || !depth) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    const size_t W = c->W, H = c->H;
    if (dstride < 2 * W || ostride < 4 * W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    DevGuard g(c->device);
    int rc;
    if ((rc = ensure(c, c->d16tmp, W * H * 2))) return rc;
    if ((rc = ensure(c, c->depth, W * H * 4))) return rc;
    HIPCHK(c, copy2d(c->d16tmp.p, 2 * W, d16, dstride, 2 * W, H, hipMemcpyHostToDevice, c->stream));
    const float* mdev = nullptr;  // NULL: the device-resident MapperL of sv_update_mapper
    if (mapper) {
        if ((rc = ensure(c, c->mapper, W * H * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->mapper.p, mapper, W * H * 8, hipMemcpyHostToDevice, c->stream));
        mdev = (const float*)c->mapper.p;
    }
    if ((rc = sv_depth_dev(c, (const int16_t*)c->d16tmp.p, mdev, Q, R, (float*)c->depth.p))) return rc;
    HIPCHK(c, copy2d(depth, ostride, c->depth.p, 4 * W, 4 * W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_world_image_dev(sv_ctx* c, const int16_t* d16, const float* mapper, const double* Q, const double* RLrect, const double* HL_root,
                       float* cart, float* cyl) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !Q || !RLrect || !HL_root || !cart || !cyl) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if (!mapper) {
        if (!c->have_mapper[0]) return fail(c, SV_ERR_NO_MAPS, "no mapper: pass one or call sv_update_mapper(ctx, 0, ...)");
        mapper = (const float*)c->mapper_lr[0].p;
    }
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    svk::WorldCoef k;
    k.q00 = Q[0]; k.q03 = Q[3]; k.q11 = Q[5]; k.q13 = Q[7]; k.q23 = Q[11]; k.q32 = Q[14]; k.q33 = Q[15];
    for (int i = 0; i < 3; i++)       // Hrect = HL_root * [RLrect^T 0; 0 1]  (SFM.cpp:1300-1302)
        for (int j = 0; j < 4; j++) {
            double a = 0.0;
            for (int t = 0; t < 4; t++) a += HL_root[i * 4 + t] * ((t < 3 && j < 3) ? RLrect[j * 3 + t] : (t == j ? 1.0 : 0.0));
            k.Hr[i * 4 + j] = a;
        }
    hipLaunchKernelGGL(svk::k_world_image, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream, d16, (const float2*)mapper, cart, cyl,
                       c->W, c->H, k);
    HIPCHK(c, hipGetLastError());
    return SV_OK;
}

int sv_world_image_host(sv_ctx* c, const int16_t* d16, size_t dstride, const float* mapper, const double* Q, const double* RLrect,
                        const double* HL_root, float* cart, size_t cart_stride, float* cyl, size_t cyl_stride) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !Q || !RLrect || !HL_root || !cart || !cyl) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    const size_t W = c->W, H = c->H;
    if (dstride < 2 * W || cart_stride < 12 * W || cyl_stride < 12 * W) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int rc;
    if ((rc = ensure(c, c->d16tmp, W * H * 2))) return rc;
    if ((rc = ensure(c, c->world_cart, W * H * 12))) return rc;
    if ((rc = ensure(c, c->world_cyl, W * H * 12))) return rc;
    HIPCHK(c, copy2d(c->d16tmp.p, 2 * W, d16, dstride, 2 * W, H, hipMemcpyHostToDevice, c->stream));
    const float* mdev = nullptr;  // NULL: the device-resident MapperL of sv_update_mapper
    if (mapper) {
        if ((rc = ensure(c, c->mapper, W * H * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->mapper.p, mapper, W * H * 8, hipMemcpyHostToDevice, c->stream));
        mdev = (const float*)c->mapper.p;
    }
    if ((rc = sv_world_image_dev(c, (const int16_t*)c->d16tmp.p, mdev, Q, RLrect, HL_root, (float*)c->world_cart.p, (float*)c->world_cyl.p)))
        return rc;
    HIPCHK(c, copy2d(cart, cart_stride, c->world_cart.p, 12 * W, 12 * W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, copy2d(cyl, cyl_stride, c->world_cyl.p, 12 * W, 12 * W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_wls_roi(const sv_ctx* c, int* roi) {
    if (!c || !roi) return SV_ERR_INVALID_ARG;
    wls_roi(c, roi);
    return SV_OK;
}

int sv_filter_wls_dev(sv_ctx* c, int n_frames, const int16_t* d16, const uint8_t* vis, const uint8_t* guide, int16_t* out16,
                      uint8_t* out8) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !guide || !out16 || n_frames < 1) return fail(c, SV_ERR_INVALID_ARG, "bad WLS argument");
    if ((vis == nullptr) != (out8 == nullptr)) return fail(c, SV_ERR_INVALID_ARG, "vis and vis_wls must both be given or both be NULL");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    return wls_dev(c, n_frames, d16, vis, guide, out16, out8);
}

int sv_filter_wls_host(sv_ctx* c, const int16_t* d16, size_t ds, const uint8_t* vis, size_t vs, const uint8_t* guide, size_t gs,
                       int16_t* out16, size_t os16, uint8_t* out8, size_t os8) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !out16) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if ((vis == nullptr) != (out8 == nullptr)) return fail(c, SV_ERR_INVALID_ARG, "vis and vis_wls must both be given or both be NULL");
    const size_t W = c->W, H = c->H;
    if (ds < 2 * W || os16 < 2 * W || (vis && (vs < W || os8 < W)) || (guide && gs < 3 * W))
        return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    if (!guide && !c->rect_l.p) return fail(c, SV_ERR_INVALID_ARG, "no guide image: pass one or call sv_compute_rgb_host first");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int rc;
    if ((rc = ensure(c, c->d16tmp, W * H * 2))) return rc;
    if ((rc = ensure(c, c->wls_o16, W * H * 2))) return rc;
    HIPCHK(c, copy2d(c->d16tmp.p, 2 * W, d16, ds, 2 * W, H, hipMemcpyHostToDevice, c->stream));
    if (vis) {
        if ((rc = ensure(c, c->wls_v, W * H))) return rc;
        if ((rc = ensure(c, c->wls_o8, W * H))) return rc;
        HIPCHK(c, copy2d(c->wls_v.p, W, vis, vs, W, H, hipMemcpyHostToDevice, c->stream));
    }
    const uint8_t* gdev = (const uint8_t*)c->rect_l.p;
    if (guide) {
        if ((rc = ensure(c, c->wls_g, W * H * 3))) return rc;
        HIPCHK(c, copy2d(c->wls_g.p, 3 * W, guide, gs, 3 * W, H, hipMemcpyHostToDevice, c->stream));
        gdev = (const uint8_t*)c->wls_g.p;
    }
    if ((rc = wls_dev(c, 1, (const int16_t*)c->d16tmp.p, vis ? (const uint8_t*)c->wls_v.p : nullptr, gdev, (int16_t*)c->wls_o16.p,
                      vis ? (uint8_t*)c->wls_o8.p : nullptr)))
        return rc;
    HIPCHK(c, copy2d(out16, os16, c->wls_o16.p, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
    if (vis) HIPCHK(c, copy2d(out8, os8, c->wls_o8.p, W, W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

// filterBLF / BLF_ORIGINAL on the device (kernels and numerics: sv_blf.hip.h).  The grid's depth
// depends on the value range of the map, so the min/max pass is followed by one small read-back.
static int blf_dev(sv_ctx* c, const int16_t* d16, double sigma_color, double sigma_space, int16_t* out16, uint8_t* out8) {
    const int W = c->W, H = c->H;
    const size_t n = (size_t)W * H;
    int rc;
    if ((rc = ensure(c, c->blf_mm, 2 * sizeof(int)))) return rc;
    int mm[2] = {32767, -32768};
    HIPCHK(c, hipMemcpyAsync(c->blf_mm.p, mm, sizeof(mm), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(svk::k_blf_minmax, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, c->stream, d16, n,
                       (int*)c->blf_mm.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(mm, c->blf_mm.p, sizeof(mm), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    svk::BlfGrid g;
    g.vmin = mm[0];
    g.sigma_color = sigma_color;
    g.sigma_space = sigma_space;
    // fastBilateral.hpp:122-124 (the depth's padding term is written with padding_xy there; both are 2)
    const size_t gh = (size_t)((double)(size_t)(H - 1) / sigma_space) + 1 + 2 * svk::BLF_PAD;
    const size_t gw = (size_t)((double)(size_t)(W - 1) / sigma_space) + 1 + 2 * svk::BLF_PAD;
    const size_t gd = (size_t)(((double)mm[1] - (double)mm[0]) / sigma_color) + 1 + 2 * svk::BLF_PAD;
    const size_t cells = gh * gw * gd;
    if (cells > ((size_t)1 << 27))  // 2 grids x 16 B x 2^27 cells = 4 GiB
        return fail(c, SV_ERR_UNSUPPORTED, "bilateral grid %zux%zux%zu too large: raise sigmaColorBLF / sigmaSpaceBLF", gh, gw, gd);
    g.gh = (int)gh; g.gw = (int)gw; g.gd = (int)gd;
    const size_t gbytes = cells * sizeof(double2);
    if ((rc = ensure(c, c->blf_a, gbytes))) return rc;
    if ((rc = ensure(c, c->blf_b, gbytes))) return rc;
    HIPCHK(c, hipMemsetAsync(c->blf_a.p, 0, gbytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->blf_b.p, 0, gbytes, c->stream));
    double2* data = (double2*)c->blf_a.p;
    double2* buffer = (double2*)c->blf_b.p;
    hipLaunchKernelGGL(svk::k_blf_splat, dim3((W + 255) / 256, H), dim3(256), 0, c->stream, d16, W, H, g, data);
    HIPCHK(c, hipGetLastError());
    const long offs[3] = {(long)(gw * gd), (long)gd, 1};
    for (int dim = 0; dim < 3; dim++)
        for (int it = 0; it < 2; it++) {
            std::swap(data, buffer);  // :156
            hipLaunchKernelGGL(svk::k_blf_blur, dim3((unsigned)((gd - 2 + 255) / 256), (unsigned)(gw - 2), (unsigned)(gh - 2)), dim3(256), 0,
                               c->stream, buffer, data, g, offs[dim]);
            HIPCHK(c, hipGetLastError());
        }
    hipLaunchKernelGGL(svk::k_blf_slice, dim3((W + 255) / 256, H), dim3(256), 0, c->stream, d16, W, H, g, data, out16);
    HIPCHK(c, hipGetLastError());
    if (out8) {  // getDisparityVis(disparity16_blf, disparity_blf, 3), StereoMatcher.cpp:97
        hipLaunchKernelGGL(svk::k_vis, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, out16, out8, n, 3);
        HIPCHK(c, hipGetLastError());
    }
    return SV_OK;
}

int sv_filter_blf_dev(sv_ctx* c, const int16_t* d16, double sigma_color, double sigma_space, int16_t* out16, uint8_t* out8) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !out16) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if (!(sigma_color > 0.0) || !(sigma_space > 0.0)) return fail(c, SV_ERR_INVALID_ARG, "sigmaColorBLF and sigmaSpaceBLF must be positive");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    return blf_dev(c, d16, sigma_color, sigma_space, out16, out8);
}

int sv_filter_blf_host(sv_ctx* c, const int16_t* d16, size_t ds, double sigma_color, double sigma_space, int16_t* out16, size_t os16,
                       uint8_t* out8, size_t os8) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!d16 || !out16) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if (!(sigma_color > 0.0) || !(sigma_space > 0.0)) return fail(c, SV_ERR_INVALID_ARG, "sigmaColorBLF and sigmaSpaceBLF must be positive");
    const size_t W = c->W, H = c->H;
    if (ds < 2 * W || os16 < 2 * W || (out8 && os8 < W)) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    int rc;
    if ((rc = ensure(c, c->d16tmp, W * H * 2))) return rc;
    if ((rc = ensure(c, c->blf_o16, W * H * 2))) return rc;
    if (out8 && (rc = ensure(c, c->blf_o8, W * H))) return rc;
    HIPCHK(c, copy2d(c->d16tmp.p, 2 * W, d16, ds, 2 * W, H, hipMemcpyHostToDevice, c->stream));
    if ((rc = blf_dev(c, (const int16_t*)c->d16tmp.p, sigma_color, sigma_space, (int16_t*)c->blf_o16.p, out8 ? (uint8_t*)c->blf_o8.p : nullptr)))
        return rc;
    HIPCHK(c, copy2d(out16, os16, c->blf_o16.p, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
    if (out8) HIPCHK(c, copy2d(out8, os8, c->blf_o8.p, W, W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

// filterBLF / BLF_CUDA semantics on the device (kernel: sv_blf.hip.h, k_blf_joint).  guide: gray u8.
static int blf_joint_dev(sv_ctx* c, const int16_t* d16, const uint8_t* guide, int ndisp, int radius, int iters, double sigma_range,
                         int16_t* out16, uint8_t* out8) {
    const int W = c->W, H = c->H;
    const size_t n = (size_t)W * H;
    int rc;
    const size_t tab_floats = 256 + (size_t)(radius + 1) * (radius + 1);
    if ((rc = ensure(c, c->blf_tab, (256 + 33 * 33) * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->blf_tmp, n * 2))) return rc;
    if (c->blf_tab_sigma != sigma_range || c->blf_tab_radius != radius) {
        // calc_color_weighted_table / calc_space_weighted_filter of OpenCV's filter, on the host
        std::vector<float> tab(tab_floats);
        for (int i = 0; i < 256; i++) tab[i] = (float)std::exp(-(double)(i * i) / (2 * sigma_range * sigma_range));
        const float dist_space = (float)radius + 1.0f;
        for (int y = 0; y <= radius; y++)
            for (int x = 0; x <= radius; x++)
                tab[256 + (size_t)y * (radius + 1) + x] = expf(-sqrtf((float)(y * y) + (float)(x * x)) / dist_space);
        HIPCHK(c, hipMemcpyAsync(c->blf_tab.p, tab.data(), tab_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // tab goes out of scope
        c->blf_tab_sigma = sigma_range;
        c->blf_tab_radius = radius;
    }
    short edge_disc = (short)(ndisp * 0.1f + 0.5);
    if (edge_disc < 1) edge_disc = 1;
    const short max_disc = (short)(ndisp * 0.2f + 0.5);
    const float* tc = (const float*)c->blf_tab.p;
    if (out16 != d16) HIPCHK(c, hipMemcpyAsync(out16, d16, n * 2, hipMemcpyDeviceToDevice, c->stream));
    const dim3 grid((unsigned)((W / 2 + 1 + 255) / 256), (unsigned)H);
    for (int it = 0; it < iters; it++)
        for (int t = 0; t < 2; t++) {
            HIPCHK(c, hipMemcpyAsync(c->blf_tmp.p, out16, n * 2, hipMemcpyDeviceToDevice, c->stream));  // the map before this half-pass
            hipLaunchKernelGGL(svk::k_blf_joint, grid, dim3(256), 0, c->stream, (const int16_t*)c->blf_tmp.p, out16, guide, tc, tc + 256,
                               W, H, t, radius, (int)edge_disc, (int)max_disc);
            HIPCHK(c, hipGetLastError());
        }
    if (out8) {  // getDisparityVis(disparity16_blf, disparity_blf, 3), StereoMatcher.cpp:113
        hipLaunchKernelGGL(svk::k_vis, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, out16, out8, n, 3);
        HIPCHK(c, hipGetLastError());
    }
    return SV_OK;
}

static int blf_joint_args(sv_ctx* c, const void* d16, const void* out16, int ndisp, int radius, int iters, double sigma_range) {
    if (!d16 || !out16) return fail(c, SV_ERR_INVALID_ARG, "NULL pointer");
    if (ndisp < 1 || radius < 1 || radius > 32 || iters < 1 || iters > 64 || !(sigma_range > 0.0))
        return fail(c, SV_ERR_INVALID_ARG, "need ndisp >= 1, radius in [1,32], iters in [1,64], sigma_range > 0");
    return SV_OK;
}

int sv_filter_blf_joint_dev(sv_ctx* c, const int16_t* d16, const uint8_t* guide_gray, int ndisp, int radius, int iters,
                            double sigma_range, int16_t* out16, uint8_t* out8) {
    if (!c) return SV_ERR_INVALID_ARG;
    int rc = blf_joint_args(c, d16, out16, ndisp, radius, iters, sigma_range);
    if (rc) return rc;
    if (!guide_gray) return fail(c, SV_ERR_INVALID_ARG, "NULL guide image");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    return blf_joint_dev(c, d16, guide_gray, ndisp, radius, iters, sigma_range, out16, out8);
}

int sv_filter_blf_joint_host(sv_ctx* c, const int16_t* d16, size_t ds, const uint8_t* guide_rgb, size_t gs, int ndisp, int radius,
                             int iters, double sigma_range, int16_t* out16, size_t os16, uint8_t* out8, size_t os8) {
    if (!c) return SV_ERR_INVALID_ARG;
    int rc = blf_joint_args(c, d16, out16, ndisp, radius, iters, sigma_range);
    if (rc) return rc;
    const size_t W = c->W, H = c->H;
    if (ds < 2 * W || os16 < 2 * W || (out8 && os8 < W) || (guide_rgb && gs < 3 * W)) return fail(c, SV_ERR_INVALID_ARG, "stride smaller than a row");
    if (!guide_rgb && !c->rect_l.p) return fail(c, SV_ERR_INVALID_ARG, "no guide image: pass one or call sv_compute_rgb_host first");
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if ((rc = ensure(c, c->d16tmp, W * H * 2))) return rc;
    if ((rc = ensure(c, c->blf_o16, W * H * 2))) return rc;
    if ((rc = ensure(c, c->blf_gray, W * H))) return rc;
    if (out8 && (rc = ensure(c, c->blf_o8, W * H))) return rc;
    HIPCHK(c, copy2d(c->d16tmp.p, 2 * W, d16, ds, 2 * W, H, hipMemcpyHostToDevice, c->stream));
    const uint8_t* rgb_dev = (const uint8_t*)c->rect_l.p;  // the rectified left image of the last sv_compute_rgb_host
    if (guide_rgb) {
        if ((rc = ensure(c, c->wls_g, W * H * 3))) return rc;
        HIPCHK(c, copy2d(c->wls_g.p, 3 * W, guide_rgb, gs, 3 * W, H, hipMemcpyHostToDevice, c->stream));
        rgb_dev = (const uint8_t*)c->wls_g.p;
    }
    // cv::cvtColor(grayL, grayL, CV_BGR2GRAY) on the RGB-ordered rectified left image (StereoMatcher.cpp:103-104)
    if ((rc = sv_gray_from_rgb_dev(c, rgb_dev, (uint8_t*)c->blf_gray.p))) return rc;
    if ((rc = blf_joint_dev(c, (const int16_t*)c->d16tmp.p, (const uint8_t*)c->blf_gray.p, ndisp, radius, iters, sigma_range,
                            (int16_t*)c->blf_o16.p, out8 ? (uint8_t*)c->blf_o8.p : nullptr)))
        return rc;
    HIPCHK(c, copy2d(out16, os16, c->blf_o16.p, 2 * W, 2 * W, H, hipMemcpyDeviceToHost, c->stream));
    if (out8) HIPCHK(c, copy2d(out8, os8, c->blf_o8.p, W, W, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SV_OK;
}

int sv_last_aggregation(const sv_ctx* c, int* n_launches, int* n_fused_sweeps) {
    if (!c || !n_launches || !n_fused_sweeps) return SV_ERR_INVALID_ARG;
    *n_launches = c->last_agg_launches;
    *n_fused_sweeps = c->last_agg_sweeps;
    return SV_OK;
}

int sv_last_sweep_waves(const sv_ctx* c) { return c ? c->last_sweep_cw : SV_ERR_INVALID_ARG; }

int sv_synchronize(sv_ctx* c) {
    if (!c) return SV_ERR_INVALID_ARG;
    DevGuard g(c->device);
    if (!g.ok) return fail(c, SV_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_sweep_error(c);
}

#ifdef SV_SWEEP_PROF
// diagnostic builds only: copies the cycle sums of the last sweep launches to the host
int sv_debug_sweep_prof(sv_ctx* c, unsigned long long* out, int n) {
    if (!c || !out || n < c->sw_prof_n) return c ? c->sw_prof_n : -1;
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(out, c->sw_prof.p, (size_t)c->sw_prof_n * 8, hipMemcpyDeviceToHost);
    return c->sw_prof_n;
}
#endif

int sv_debug_sweep_order(sv_ctx* c, int enable, int* launches, int* violations, long long* worst_lead_ticks, long long* worst_lead_prev_ticks) {
    if (!c) return SV_ERR_INVALID_ARG;
    DevGuard g(c->device);
    int rc = sweep_order_collect(c);   // the launch still pending, if any
    if (rc) return rc;
    if (launches) *launches = c->order_launches;
    if (violations) *violations = c->order_violations;
    if (worst_lead_ticks) *worst_lead_ticks = c->order_worst;
    if (worst_lead_prev_ticks) *worst_lead_prev_ticks = c->order_worst_prev;
    c->order_launches = c->order_violations = 0;
    c->order_worst = c->order_worst_prev = 0;
    c->order_check = enable != 0;
    return SV_OK;
}

int sv_debug_selftest_min3(sv_ctx* c, unsigned long long* mismatches) {
    if (!c || !mismatches) return SV_ERR_INVALID_ARG;
    DevGuard g(c->device);
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 8));
    hipError_t e = hipMemsetAsync(d, 0, 8, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(svk::k_min3_selftest, dim3(0x7C00), dim3(256), 0, c->stream, d, 0x7C00u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return SV_OK;
}

int sv_set_debug(sv_ctx* c, int enable) {
    if (!c) return SV_ERR_INVALID_ARG;
    c->debug = enable == 1;
    c->debug_mode = enable;
    c->force_fuse = enable == 2;
    c->sweep_mode = enable == 3 ? 1 : (enable == 4 ? -1 : 0);
    c->fanout_mode = enable == 5 ? 1 : (enable == 4 || enable == 2 ? -1 : 0);
    return SV_OK;
}

int sv_debug_read_volume_host(sv_ctx* c, int which, int16_t* vol, size_t n_elems) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (which == 1 && !c->debug) return fail(c, SV_ERR_INVALID_ARG, "S is only materialised in debug mode (sv_set_debug)");
    const size_t nvol = (size_t)geo_w1(c->W, c->prm) * c->H * c->prm.num_disparities;
    if (!vol || n_elems != nvol || which < 0 || which > 1) return fail(c, SV_ERR_INVALID_ARG, "bad volume request");
    DevGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(vol, which ? c->S.p : c->C.p, nvol * 2, hipMemcpyDeviceToHost));
    return SV_OK;
}

int sv_debug_read_plane_host(sv_ctx* c, int which, int16_t* plane) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!plane || which < 0 || which > 3) return fail(c, SV_ERR_INVALID_ARG, "bad plane request");
    DevGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t W = c->W, H = c->H;
    if (which == 3) {
        HIPCHK(c, hipMemcpy(plane, c->disp_lr.p, W * H * 2, hipMemcpyDeviceToHost));
        return SV_OK;
    }
    // one int16 field of the 8-byte WTA records; columns [0, D) are never written by the fused path
    HIPCHK(c, hipMemcpy2D(plane, 2, (const char*)c->rec.p + 2 * which, 8, 2, W * H, hipMemcpyDeviceToHost));
    const int16_t inval[3] = {(int16_t)geo_inv(c->prm), -1, 32767};
    const int col0 = geo_col0(c->W, c->prm), colN = geo_colN(c->W, c->prm);
    for (size_t y = 0; y < H; y++)
        for (int x = 0; x < (int)W; x++)
            if (x < col0 || x >= colN) plane[y * W + x] = inval[which];
    return SV_OK;
}

int sv_enable_timing(sv_ctx* c, int enable) {
    if (!c) return SV_ERR_INVALID_ARG;
    c->timing = enable != 0;
    c->ev_calls = 0;
    return SV_OK;
}

int sv_get_timings(sv_ctx* c, float* ms, int n) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!ms || n < SV_N_TIMINGS) return fail(c, SV_ERR_INVALID_ARG, "need room for %d timings", SV_N_TIMINGS);
    if (c->ev_calls <= 0) return fail(c, SV_ERR_INVALID_ARG, "no timed call yet (sv_enable_timing, then compute)");
    DevGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // marks of one call: 0 start, 1 after stats, 2 after cost, 3 after agg, 4 after wta, 5 after tail.
    // Mean over the calls still in the ring (at most EV_RING most recent ones).
    const long nset = c->ev_calls < sv_ctx::EV_RING ? c->ev_calls : sv_ctx::EV_RING;
    double acc[SV_N_TIMINGS] = {0};
    for (long k = 0; k < nset; k++) {
        hipEvent_t* evs = c->ev[(c->ev_calls - 1 - k) % sv_ctx::EV_RING];
        float t;
        for (int i = 0; i < 5; i++) {
            HIPCHK(c, hipEventElapsedTime(&t, evs[i], evs[i + 1]));
            acc[i] += t;
        }
        HIPCHK(c, hipEventElapsedTime(&t, evs[0], evs[5]));
        acc[5] += t;
    }
    for (int i = 0; i < SV_N_TIMINGS; i++) ms[i] = (float)(acc[i] / (double)nset);
    return SV_OK;
}

int sv_get_aggregation_timings(sv_ctx* c, float* ms, int n, int* n_launches) {
    if (!c) return SV_ERR_INVALID_ARG;
    if (!ms || !n_launches || n < 1) return fail(c, SV_ERR_INVALID_ARG, "bad argument");
    if (c->ev_calls <= 0) return fail(c, SV_ERR_INVALID_ARG, "no timed call yet (sv_enable_timing, then compute)");
    DevGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // launches of the most recent call; the mean runs over the calls in the ring that launched the same number
    const int nl = c->ev_agg_n[(c->ev_calls - 1) % sv_ctx::EV_RING];
    *n_launches = nl;
    const long nset = c->ev_calls < sv_ctx::EV_RING ? c->ev_calls : sv_ctx::EV_RING;
    double acc[8] = {0};
    long used = 0;
    for (long k = 0; k < nset; k++) {
        const long slot = (c->ev_calls - 1 - k) % sv_ctx::EV_RING;
        if (c->ev_agg_n[slot] != nl) continue;
        used++;
        for (int i = 0; i < nl; i++) {
            float t;
            HIPCHK(c, hipEventElapsedTime(&t, i == 0 ? c->ev[slot][2] : c->ev_agg[slot][i - 1], c->ev_agg[slot][i]));
            acc[i] += t;
        }
    }
    for (int i = 0; i < n; i++) ms[i] = i < nl && used ? (float)(acc[i] / (double)used) : 0.0f;
    return SV_OK;
}

int64_t sv_algorithmic_bytes(const sv_ctx* c, int which) {
    if (!c) return -1;
    const int64_t W = c->W, H = c->H, D = c->prm.num_disparities;
    const int64_t V = (int64_t)geo_w1(c->W, c->prm) * H * D * 2;
    int64_t P = 0;
    for (int i = 0; i < 8; i++) P += (c->prm.path_mask >> i) & 1;
    switch (which) {
        case 0: return V + 2 * W * H;
        case 1: return 3 * P * V;
        case 2: return V + 6 * W * H;
        case 3: return 8 * W * H;
        case 4: return (3 * P + 2) * V + 16 * W * H;
        default: return -1;
    }
}


// ---------------------------------------------------------------------------------------------
// In-process multi-device dispatch: stereo pairs are independent, so a host that
// serves many camera streams deals chunks of pairs to one context per device, one host thread and
// one stream per context, no collective, no traffic between devices; every pair has its own
// output pointer, so results land in submission order by construction.
// ---------------------------------------------------------------------------------------------
struct sv_pool {
    std::vector<sv_ctx*> ctx;
    std::vector<hipStream_t> stream;
    int W = 0, H = 0, batch = 0;
    char err[512] = {0};
};

int sv_pool_plan(int n_frames, int n_contexts, int batch, int frame, int* context_index, int* chunk_index) {
    if (n_frames < 0 || n_contexts < 1 || batch < 1 || frame < 0 || frame >= n_frames || !context_index || !chunk_index)
        return SV_ERR_INVALID_ARG;
    const int chunk = frame / batch;      // chunks of `batch` consecutive frames ...
    *chunk_index = chunk;
    *context_index = chunk % n_contexts;  // ... dealt round-robin to the contexts
    return SV_OK;
}

const char* sv_pool_last_error(const sv_pool* p) { return p ? p->err : k_noctx; }

int sv_pool_destroy(sv_pool* p) {
    if (!p) return SV_ERR_INVALID_ARG;
    for (size_t i = 0; i < p->ctx.size(); i++) {
        const int dev = p->ctx[i] ? p->ctx[i]->device : -1;
        if (p->ctx[i]) sv_destroy(p->ctx[i]);
        if (p->stream[i] && dev >= 0) {
            DevGuard g(dev);
            (void)hipStreamDestroy(p->stream[i]);
        }
    }
    delete p;
    return SV_OK;
}

int sv_pool_create(sv_pool** out, const int* devices, int n_devices, int width, int height, const sv_params* params, int batch) {
    if (!out) return SV_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64 || batch < 1) return SV_ERR_INVALID_ARG;
    sv_pool* p = new (std::nothrow) sv_pool();
    if (!p) return SV_ERR_NOMEM;
    *out = p;  // returned even on failure: sv_pool_last_error, then sv_pool_destroy
    p->W = width; p->H = height; p->batch = batch;
    p->ctx.assign(n_devices, nullptr);
    p->stream.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; i++) {
        int rc = sv_create(&p->ctx[i], devices[i], width, height, params);
        if (rc == SV_OK) rc = sv_set_batch(p->ctx[i], batch);
        if (rc == SV_OK) {
            DevGuard g(devices[i]);
            // a stream of its own: two contexts on one device (or the caller's other work) overlap instead of queueing on stream 0
            if (hipStreamCreateWithFlags(&p->stream[i], hipStreamNonBlocking) != hipSuccess) rc = SV_ERR_HIP;
            else rc = sv_set_stream(p->ctx[i], p->stream[i]);
        }
        if (rc != SV_OK) {
            snprintf(p->err, sizeof(p->err), "context %d (device %d): %s", i, devices[i], p->ctx[i] ? sv_last_error(p->ctx[i]) : "sv_create failed");
            return rc;
        }
    }
    return SV_OK;
}

int sv_pool_size(const sv_pool* p) { return p ? (int)p->ctx.size() : SV_ERR_INVALID_ARG; }

int sv_pool_compute_gray_host(sv_pool* p, int n_frames, const uint8_t* const* left, size_t ls, const uint8_t* const* right, size_t rs,
                              int16_t* const* disp16, size_t ds) {
    if (!p) return SV_ERR_INVALID_ARG;
    if (n_frames < 0 || (n_frames > 0 && (!left || !right || !disp16))) { snprintf(p->err, sizeof(p->err), "bad argument"); return SV_ERR_INVALID_ARG; }
    const int n_ctx = (int)p->ctx.size(), batch = p->batch;
    const int n_chunks = (n_frames + batch - 1) / batch;
    std::vector<int> rcs(n_ctx, SV_OK);
    auto work = [&](int k) {
        sv_ctx* c = p->ctx[k];
        int tickets[SV_BATCH_SLOTS];
        int in_flight = 0, head = 0;   // ring of tickets, oldest at head
        for (int chunk = k; chunk < n_chunks && rcs[k] == SV_OK; chunk += n_ctx) {   // sv_pool_plan: chunk -> context chunk % n_ctx
            if (in_flight == SV_BATCH_SLOTS) {
                rcs[k] = sv_wait_batch_host(c, tickets[head]);
                head = (head + 1) % SV_BATCH_SLOTS;
                in_flight--;
                if (rcs[k] != SV_OK) break;
            }
            const int f0 = chunk * batch, n = std::min(batch, n_frames - f0);
            int t = -1;
            rcs[k] = sv_submit_gray_batch_host(c, n, left + f0, ls, right + f0, rs, disp16 + f0, ds, &t);
            if (rcs[k] != SV_OK) {
                // a submit that failed half-way may have queued copies or kernels without a ticket to wait on: nothing of
                // this context may still read the caller's buffers or write disp16 when the call returns
                (void)hipSetDevice(c->device);
                if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
                (void)hipStreamSynchronize(c->stream);
                if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
                break;
            }
            tickets[(head + in_flight) % SV_BATCH_SLOTS] = t;
            in_flight++;
        }
        while (in_flight > 0) {   // also after an error: nothing of this context may stay in flight when the call returns
            const int rc = sv_wait_batch_host(c, tickets[head]);
            if (rcs[k] == SV_OK) rcs[k] = rc;
            head = (head + 1) % SV_BATCH_SLOTS;
            in_flight--;
        }
    };
    std::vector<std::thread> threads;
    std::vector<int> inline_ctx;   // contexts whose thread could not be started: served by the calling thread afterwards
    for (int k = 1; k < n_ctx; k++) {
        try {   // std::thread's constructor throws std::system_error when the process is out of threads: not through a C ABI
            threads.emplace_back(work, k);
        } catch (...) {
            inline_ctx.push_back(k);
        }
    }
    work(0);   // the calling thread serves context 0
    for (int k : inline_ctx) work(k);
    for (auto& t : threads) t.join();
    for (int k = 0; k < n_ctx; k++)
        if (rcs[k] != SV_OK) {
            snprintf(p->err, sizeof(p->err), "context %d (device %d): %s", k, p->ctx[k]->device, sv_last_error(p->ctx[k]));
            return rcs[k];
        }
    return SV_OK;
}

}  // extern "C"
