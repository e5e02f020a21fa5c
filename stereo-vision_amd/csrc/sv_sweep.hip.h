// sv_sweep.hip.h — fused SGM sweep: the four paths of one sweep in ONE pass over the cost volume.
//
// Result of cost_aggregation_lr + _ud_lr + _ud + _rl_ud (sgbm_cuda/aggregation.cu:60-543) for the
// first sweep, and of the mirrored four paths of OpenCV MODE_HH's second sweep (what matchSGBM asks
// for, StereoMatcher.cpp:409) + get_disparity (aggregation.cu:629-689) for the second.  The
// per-path kernels (k_agg) read C and read-modify-write S once per path: 22 volume passes for
// eight paths.  Here a sweep reads C once and writes (sweep 1) or reads (sweep 2) S once: 4 passes.
//
// How: the predecessors of pixel (x, y) in a sweep are (x-1, y), (x-1, y-1), (x, y-1), (x+1, y-1).
// Give every image row its own 16-lane group that walks the row left to right, and let row y run
// two pixels behind row y-1: at step t row y is at x = t - 2y, and everything it needs from the
// row above was produced 1 (down-left), 2 (down) and 3 (down-right) steps earlier.  The image is a
// systolic array of rows; the horizontal path lives in the row's own registers, the other three
// arrive from the row above through an LDS mailbox (one record per row and step, double-buffered,
// one workgroup barrier per step).  A workgroup holds 60 consecutive rows (15 waves) plus one
// service wave that carries the records across the band boundary: it exports the last row's
// records to HBM and imports the previous band's, ordered by a progress counter per band
// (write-through sc1 stores, drained, then an sc1 counter store; sc1 polls and loads on the other
// side: MI355X_MICROARCH.md "Valid forms", first row of the table).  Bands of one frame therefore
// run as a pipeline, 2*60 steps apart, and a workgroup streams its band of frame after frame
// without draining, so the 2-steps-per-row skew is paid once per launch, not once per frame.
// Dispatch order: band b waits for band b-1 of its group, which has the next lower blockIdx.  The host sizes
// a launch to what the chip holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor; more frames = more
// frames per group or further launches), so alone on the GPU every workgroup is resident and the order in
// which they start does not matter.  When the launch shares the GPU (another context's or process's
// kernels) or is oversubscribed on purpose (SV_SWEEP_NCU, tests), what keeps it live is the dispatcher's
// rule for a 1-D grid: workgroup i goes to XCD i mod 8, and every XCD starts its workgroups in blockIdx
// order.  Then the unfinished workgroup with the lowest blockIdx is always resident (it was first in its
// XCD's queue) and waits for nobody that is not finished, so the launch always advances - although a
// workgroup may start 100+ us before workgroup blockIdx - 1, which sits on another XCD (measured).  The
// rule is observed, not an architectural promise: tests/test_gpu_sweep.py::test_sweep_grid_exceeds_residency
// runs a grid of four times the capacity and checks it with sv_debug_sweep_order (start stamps per workgroup).
// Every wait is bounded in time (SW_WAIT_TICKS); a timeout sets the host-visible error word, publishes the
// band as finished so that the launch drains, sv_synchronize / the *_host entry points report it and the
// context goes over to the per-path kernels.
// The second sweep is the same kernel on coordinates mirrored in x and y; its finished S stays in
// registers and goes straight into winner-take-all.
#pragma once
#include "sv_kernels.hip.h"

namespace svk {

// pixels in the import / export rings: a record is written <= 3 steps before it is read, so 4 are enough; the 9-wave
// shape for D > 128 takes 4 (its 32 rows of 1.6 KB records leave no room for more), the others 8
template <int CW>
__host__ __device__ constexpr int sweep_ring() { return CW == 8 ? 4 : 8; }
constexpr int SW_LEAD = 32;   // loop steps before stream pixel 0 (service pipeline warm-up), multiple of 4
#ifndef SV_SW_DEPTH
#define SV_SW_DEPTH 2
#endif
static_assert(SV_SW_DEPTH == 2 || SV_SW_DEPTH == 4, "even, and the import runs at most six pixels ahead of the counter check");
constexpr int SW_DEPTH = SV_SW_DEPTH;   // global-load pipeline depth in steps (even: the record slots alternate with the step's parity).
                                        // A step takes ~2 us, longer than an HBM access: two steps of loads in flight are enough,
                                        // and each step of depth costs 2N registers of a 128-register budget
// Bound of every wait on a neighbouring band, in ticks of the constant 100 MHz counter (s_memrealtime): 2 s of
// time actually spent polling.  One poll interval counts for at most SW_WAIT_TICK_CAP ticks, so a pause of the
// whole queue (another process's kernels time-sliced onto the GPU) does not run the bound out.
constexpr unsigned long long SW_WAIT_TICKS = 200000000ull;
constexpr unsigned long long SW_WAIT_TICK_CAP = 1000ull;

struct SweepArgs {
    const uint32_t* C;             // [n_frames][H][W1][D/2] packed u16 pairs
    uint32_t* S;                   // same shape: written by sweep 1, read by sweep 2
    uint2* rec;                    // sweep 2: WTA records [n_frames][H][W]
    unsigned long long* handoff;   // [n_groups][n_bands - 1][ho_px + 1][REC / 2]: band-boundary records (+1 dummy)
    unsigned int* progress;        // [n_groups][n_bands]: stream pixels exported so far (zeroed before the launch)
    unsigned int* error;           // host-visible: set when a bounded wait ran out
    uint32_t* sink;                // [gridDim.x * blockDim.x * sweep_sink_words<N>()] u32: target of the stores of inactive steps
    int W1, H, W;
    int col0, min_disp;            // first computable column; minDisparity (EXT): stored disparity = (index + min_disp) * 16 + sub-pixel
    uint32_t P1P1, P2P2;
    int uniq;
    int n_frames, n_groups, n_bands;
    int fpg;                       // frames per group (the last group may hold fewer)
    int ho_px;                     // handoff pixels per (group, band): frames-per-group * W1, rounded up
    unsigned long long* starts;    // dispatch-order check (sv_debug_sweep_order): [gridDim.x] s_memrealtime at workgroup start, or nullptr
    unsigned long long* prof;      // diagnostic builds (-DSV_SWEEP_PROF): [gridDim.x][waves][4] cycle sums, else unused
};

template <int N>
__host__ __device__ constexpr int sweep_sink_words() { return N < 2 ? 2 : N; }   // per thread: N words of S, or one 8-byte WTA record
template <int N>
__host__ __device__ constexpr int sweep_rec_words() { return 3 * 16 * N + 16; }
// CW = compute waves per workgroup (4 image rows each) + one service wave
// Sweep 2 hands every finished pixel to the service wave for winner-take-all through a per-row LDS slot, double-buffered
// by the step's parity: the 32 N halfwords of S in disparity order, then the 16 lanes' local first-minimum keys
// (S << 16 | d), then 4 words of padding: a row stride of 16 N + 20 words is an odd multiple of 4, so the 16-byte reads of
// the service wave (one lane per row, all lanes at the same offset of their row) spread over all banks.
template <int N>
__host__ __device__ constexpr int sweep_wta_words() { return 16 * N + 20; }
template <int N, int CW, bool SECOND>
__host__ __device__ constexpr size_t sweep_lds_bytes() {
    return sizeof(uint32_t) * ((size_t)(2 * (4 * CW + 1) + 2 * sweep_ring<CW>()) * sweep_rec_words<N>() + 4 +
                               (SECOND ? (size_t)2 * 4 * CW * sweep_wta_words<N>() : 0));
}

// One SGM path update of one pixel: r = previous L along the path (0 with dd = P2 when the path
// starts here), c = the pixel's costs.  Same arithmetic as k_agg's step:
//   L = C + min(Lp(d), Lp(d-1) + P1, Lp(d+1) + P1, dd) - dd  =  C - max(dd - a, 0).
// The neighbour lanes' registers arrive through DPP row shifts folded into an OR (v_or_b32_dpp with
// bound_ctrl: the lane with no neighbour reads 0) with a per-lane constant: "no such disparity" =
// SW_BIG in lane 0 (big15: lane 15) of the row, 0 elsewhere.  One instruction per shift, no landing
// register to initialise or keep.
// Packed three-input minimum of u16 pairs: v_pk_minimum3_f16 (new on gfx950).  Non-negative finite binary16 patterns
// (0 .. 0x7BFF) order like the integers they spell, denormals included under the kernel's default f16 denormal mode, and a minimum
// returns one of its inputs bit for bit; scratch/ubench/pkmin3.hip checks every pair of [0, 0x7C00)^2 on the GPU and times it
// (one issue slot, like v_pk_min_u16).  The host runs the sweeps only where every value that meets one of these stays below
// 0x7C00 (sweep_f16_ok: largest cost + P1 <= 0x7BFF); the thin slice of parameter sets above that takes the per-path kernels.
constexpr uint32_t SW_BIG = 0x7BFF7BFFu;   // "no such disparity": never below a real value, and still a finite binary16
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
#ifdef SV_SWEEP_U16MIN
    return pk_min(pk_min(a, b), c);
#else
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}
// Self-test of that claim with this library's own build of pk_min3 (sv_debug_selftest_min3): block a, thread-strided b over
// [0, lim)^2, a hashed third operand, both halves of the packed word, against the integer minimum.
__global__ __launch_bounds__(256) void k_min3_selftest(unsigned long long* bad, uint32_t lim) {
    const uint32_t a = blockIdx.x;
    unsigned long long n = 0;
    for (uint32_t b = threadIdx.x; b < lim; b += blockDim.x) {
        const uint32_t c = ((a * 2654435761u) ^ (b * 40503u) ^ (b >> 3)) % lim, c2 = (c * 7u + a) % lim;
        const uint32_t got = pk_min3(a | (b << 16), b | (c << 16), c | (c2 << 16));
        n += got != (min(min(a, b), c) | (min(min(b, c), c2) << 16));
    }
    if (n) atomicAdd(bad, n);
}
template <int N>
__device__ __forceinline__ void sweep_path(const uint32_t (&c)[N], const uint32_t (&r)[N], uint32_t dd, uint32_t P1P1,
                                           uint32_t (&L)[N], uint32_t& mn_out, uint32_t big0, uint32_t big15) {
    // P1 goes onto the previous L BEFORE the neighbours are fetched, so that min(Lp(d), Lp(d-1) + P1, Lp(d+1) + P1) is one
    // three-input minimum per register: 4 instructions per register (add, min3, saturating subtract, subtract) instead of 6.
    // Plain 32-bit add / subtract where no half can carry or borrow: every L <= 0x7BFF - P1; C >= P2 >= max(dd - t, 0).
    uint32_t rp[N];
#pragma unroll
    for (int k = 0; k < N; k++) rp[k] = r[k] + P1P1;
    const uint32_t prevHi = big0 | dpp<DPP_ROW_SHR1>(0u, rp[N - 1]);
    const uint32_t nextLo = big15 | dpp<DPP_ROW_SHL1>(0u, rp[0]);
    uint32_t m[N + 1];
    m[0] = __builtin_amdgcn_alignbit(rp[0], prevHi, 16);
#pragma unroll
    for (int k = 1; k < N; k++) m[k] = __builtin_amdgcn_alignbit(rp[k], rp[k - 1], 16);
    m[N] = __builtin_amdgcn_alignbit(nextLo, rp[N - 1], 16);
#pragma unroll
    for (int k = 0; k < N; k++) L[k] = c[k] - pk_sub_sat(dd, pk_min3(m[k], m[k + 1], r[k]));
    uint32_t mn = (N & 1) ? L[N - 1] : pk_min(L[N - 2], L[N - 1]);
#pragma unroll
    for (int k = 0; k + 1 < N - 1; k += 2) mn = pk_min3(mn, L[k], L[k + 1]);
    mn_out = mn;  // the minima of the even and of the odd disparities of this lane; sweep_lane_min joins the halves
}
// the four paths' lane minima, in both halves of the word; together, so that the four two-instruction chains interleave
__device__ __forceinline__ void sweep_lane_min4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    const uint32_t ta = __builtin_amdgcn_alignbit(a, a, 16), tb = __builtin_amdgcn_alignbit(b, b, 16);
    const uint32_t tc = __builtin_amdgcn_alignbit(c, c, 16), td = __builtin_amdgcn_alignbit(d, d, 16);
    a = pk_min(a, ta); b = pk_min(b, tb); c = pk_min(c, tc); d = pk_min(d, td);
}

#ifdef SV_SWEEP_PROF
// Diagnostic build only: s_memtime stamps around the phases of a step (MI355X_MICROARCH.md, DVFS
// give-back item 6: never quote this build's run time, read its shares).
#define SW_STAMP(v)                                   \
    __builtin_amdgcn_sched_barrier(0);                \
    unsigned long long v;                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);
#define SW_PROF(...) __VA_ARGS__
#else
#define SW_STAMP(v)
#define SW_PROF(...)
#endif

#ifndef SW_ABL
#define SW_ABL 0   // diagnostic builds: bit 0 no service-wave WTA, bit 1 sweep 1 stores S to the sink, bit 2 one path skipped (wrong results)
#endif
#ifndef SW_EARLY_FENCE
#define SW_EARLY_FENCE __builtin_amdgcn_sched_barrier(0);
#endif
__device__ __forceinline__ void sweep_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier.  Not __syncthreads(): that would
    // also drain vmcnt and with it the global-load pipeline.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int N, bool SECOND, int CW>
__global__ __launch_bounds__((CW + 1) * 64) void k_sweep(SweepArgs a) {
    constexpr int SW_ROWS = 4 * CW, NTHR = (CW + 1) * 64, SW_RING = sweep_ring<CW>();
    constexpr int PXW = 16 * N, REC = sweep_rec_words<N>(), RECQ = REC / 2;
    extern __shared__ __align__(16) uint32_t sw_lds[];
    uint32_t* const slot = sw_lds;                              // [2][SW_ROWS + 1][REC]: slot[p][r] is read by row r
    uint32_t* const imp = slot + 2 * (SW_ROWS + 1) * REC;       // [SW_RING][REC]: records of the previous band's last row
    uint32_t* const expo = imp + SW_RING * REC;                 // [SW_RING][REC]: records of this band's last row
    uint32_t* const abortp = expo + SW_RING * REC;              // [0]: set by the service wave when a bounded wait ran out

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane & 15;
    const int band = blockIdx.x % a.n_bands, group = blockIdx.x / a.n_bands;
    const int f0 = group * a.fpg;                          // this group's frames: f0 .. f0 + nfr - 1 (a contiguous block)
    const int nfr = min(a.fpg, a.n_frames - f0);
    const int W1 = a.W1, H = a.H;
    const int total_px = nfr * W1;  // length of the pixel stream of every row of this group
    const bool compute = wave < CW;
    const int r = compute ? wave * 4 + (lane >> 4) : 0;  // local row
    const int yl = band * SW_ROWS + r;                    // row in sweep order
    const bool rowok = compute && yl < H;
    const int yp = SECOND ? H - 1 - min(yl, H - 1) : min(yl, H - 1);  // image row
    // Addresses: 32-bit word offsets from the group's first frame (the host keeps a group's block of
    // C below 4 GiB), so a load is one saddr + voffset instruction and a step costs a few VALU ops.
    const uint32_t frame_words = (uint32_t)H * W1 * PXW;
    const uint32_t* const Cg = a.C + (size_t)f0 * frame_words;
    uint32_t* const Sg = a.S + (size_t)f0 * frame_words;
    uint2* const recg = a.rec + (size_t)f0 * H * a.W;
    const uint32_t rowoff = (uint32_t)yp * W1 * PXW + q * N;
    const uint32_t P1P1 = a.P1P1, P2P2 = a.P2P2;
    uint32_t* const sinkp = a.sink + ((size_t)blockIdx.x * NTHR + tid) * sweep_sink_words<N>();

    static_assert(sweep_lds_bytes<N, CW, SECOND>() <= 160 * 1024, "LDS of a CU");
    // sweep 2: winner-take-all slots [2][SW_ROWS][WSTR] (sweep_wta_words): written by a row's lanes, read by lane `row` of the service wave
    constexpr int WSTR = sweep_wta_words<N>();
    uint32_t* const wta0 = abortp + 4;
    uint32_t* const wrow = wta0 + (compute ? r : min(lane, SW_ROWS - 1)) * WSTR;   // this row's slot of parity 0
    if (SECOND)
        for (int i = tid; i < 2 * SW_ROWS * WSTR; i += NTHR) wta0[i] = 0u;        // well-formed keys (best = 0) before the first real ones
    if (tid == 0) (*abortp) = 0;
    if (a.starts != nullptr && tid == 0) a.starts[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // Band 0 has no row above: its import ring is never written, so fill it once with the record
    // of a path start (L = 0, delta = P2) and the first image row needs no special case.
    if (band == 0)
        for (int i = tid; i < SW_RING * REC; i += NTHR) imp[i] = (i % REC) >= 3 * PXW ? P2P2 : 0u;
    __syncthreads();

    // ---- compute rows.  One cursor per lane, for the loads, SW_DEPTH steps ahead of the pixel being computed:
    // lx = its position in the frame row (sweep order), loff = the word offset of that pixel.  A step moves it by one
    // pixel (dstep words) or, at the end of a frame row, to the first pixel of the same row of the next frame (jump
    // words); before the stream begins the cursor walks the frames "before frame 0" in modular arithmetic, so it
    // arrives at pixel 0 of frame 0 by its ordinary wrap, and whether a pixel belongs to the stream at all is one
    // unsigned compare of its stream index (loads outside the stream read the row's first pixel instead: valid, unused).
    const int ulane = 2 * r + 2;                                   // stream index of a step's pixel = step - ulane
    const uint32_t total_eff = rowok ? (uint32_t)total_px : 0u;
    const int u_first = -SW_LEAD - ulane;                          // stream index of the first load (< 0)
    const int back = (-u_first + W1 - 1) / W1;                     // frames "before frame 0" it starts in
    int lx = u_first + back * W1;                                  // in [0, W1)
    const uint32_t dstep = SECOND ? (uint32_t)-PXW : (uint32_t)PXW;
    const uint32_t jump = frame_words - (uint32_t)(W1 - 1) * dstep;
    const uint32_t row0off = rowoff + (SECOND ? (uint32_t)(W1 - 1) * PXW : 0u);   // first pixel of the row in sweep order
    uint32_t loff = row0off - (uint32_t)back * frame_words + (uint32_t)lx * dstep;
    // the pixel being computed is SW_DEPTH behind the cursor: it is the first / last of its frame row when lx is
    const int lx_first = SW_DEPTH % W1, lx_last = (W1 - 1 + SW_DEPTH) % W1;
    // own horizontal L; records of the row above waiting for their turn: the "down" part is used one
    // step after it was read, the "down-right" part two steps after (two registers sets, alternating)
    uint32_t r0[N], dn[N], dr[2][N];
    uint32_t dd0 = P2P2, dn_dd = P2P2, dr_dd[2] = {P2P2, P2P2};
#pragma unroll
    for (int k = 0; k < N; k++) r0[k] = dn[k] = dr[0][k] = dr[1][k] = 0;
    const UniqK uqk = make_uniq(a.uniq);
    uint32_t sp[N];   // sweep 2: finished S of the previous step's pixel (goes to the service wave's slot during this step)
#pragma unroll
    for (int k = 0; k < N; k++) sp[k] = 0;
    const uint32_t big0 = q == 0 ? SW_BIG : 0u, big15 = q == 15 ? SW_BIG : 0u;  // see sweep_path
    uint32_t cb[SW_DEPTH][N], sb[SW_DEPTH][N], offq[SW_DEPTH];
    const uint32_t frame_px = (uint32_t)H * a.W;
    bool actq[SW_DEPTH];

    auto issue = [&](uint32_t (&c)[N], uint32_t (&s)[N], uint32_t& off, bool& act, int step) {
        act = (uint32_t)(step - ulane) < total_eff;
        off = act ? loff : row0off;
        load_regs<N>(Cg + off, c);
        if (SECOND) load_regs<N>(((SW_ABL & 16) ? const_cast<uint32_t*>(Cg) : Sg) + off, s);
        lx++;
        const bool wrap = lx == W1;
        lx = wrap ? 0 : lx;
        loff += wrap ? jump : dstep;
    };

    // ---- service wave state
    const bool do_import = !compute && band > 0, do_export = !compute && band < a.n_bands - 1;
    unsigned long long* const ho_in = a.handoff + ((size_t)group * (a.n_bands - 1) + max(band - 1, 0)) * (size_t)(a.ho_px + 1) * RECQ;
    unsigned long long* const ho_out = a.handoff + ((size_t)group * (a.n_bands - 1) + max(min(band, a.n_bands - 2), 0)) * (size_t)(a.ho_px + 1) * RECQ;
    unsigned int* const flag_in = a.progress + (size_t)group * a.n_bands + max(band - 1, 0);
    unsigned int* const flag_out = a.progress + (size_t)group * a.n_bands + band;
    constexpr int NQ = (RECQ + 63) / 64;  // 8-byte words of a record per lane of the service wave (2 up to D = 128, 4 at 256)
    unsigned long long ib[SW_DEPTH][NQ];
    int li[NQ];                     // this lane's 8-byte words of a record (past the end: the last word again)
    unsigned long long startq[NQ];  // ... and what they hold in the record of a path start (L = 0, delta = P2)
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        li[i] = min(lane + 64 * i, RECQ - 1);
        startq[i] = 2 * li[i] >= 3 * PXW ? (((unsigned long long)P2P2 << 32) | P2P2) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < SW_DEPTH; j++)
#pragma unroll
        for (int i = 0; i < NQ; i++) ib[j][i] = 0;

    // winner-take-all (sweep 2): lane `row` of the service wave finishes the pixel its row completed two steps earlier
    const int wrow_i = min(lane, SW_ROWS - 1), wyl = band * SW_ROWS + wrow_i;
    const bool wrowok = lane < SW_ROWS && wyl < H;
    int wx = -SW_LEAD - 2 * wrow_i - 4;              // that pixel's position in its frame at step -SW_LEAD (< 0: not started)
    int wfi = 0;
    uint32_t wrec = (uint32_t)(H - 1 - min(wyl, H - 1)) * a.W + a.col0;   // record index of the row's first pixel in frame wfi

    static_assert(N >= 1 && (N <= 4 || (N <= 8 && CW <= 8)),
                  "k_sweep keeps four paths of D = 32*N disparities in registers: D <= 128 with 16 waves per workgroup (128 "
                  "VGPRs each), D <= 256 with 9 (170 VGPRs each; 132 used)");
    const int tau_end = total_px + 2 * SW_ROWS + 4;
    // Two loops with the same trip count and one barrier per step each: the compute waves and the
    // service wave never share a basic block, so each body is straight-line code for the scheduler.
    if (compute) {
#pragma unroll
        for (int j = 0; j < SW_DEPTH; j++) {
            // sweep 1: the loop is entered with the sequence of vector memory operations in flight that its back edge
            // carries (a store between the loads), so that the compiler's wait counts stay exact
            if (!SECOND && j > 0) store_regs<N>(sinkp, r0);
            issue(cb[j], sb[j], offq[j], actq[j], -SW_LEAD + j);
            __builtin_amdgcn_sched_barrier(0);   // the loads stay in the order the loop issues them in (exact vmcnt counts)
        }
        SW_PROF(unsigned long long pacc[4] = {0, 0, 0, 0};)
        for (int tau = -SW_LEAD; tau < tau_end; tau += SW_DEPTH) {
            if ((tau & 15) == 0 && (*abortp) != 0) break;
#pragma unroll
            for (int j = 0; j < SW_DEPTH; j++) {
                SW_STAMP(T0)
                const int u = tau + j - 2 * r - 2;  // stream pixel of this row at this step; (tau + j) & 1 == j & 1
                const bool active = actq[j];
                // record of the row above: pixel u + 1 of its stream
                const uint32_t* src = (r == 0) ? imp + ((u + 1) & (SW_RING - 1)) * REC
                                               : slot + (((j + 1) & 1) * (SW_ROWS + 1) + r) * REC;
                uint32_t in_dl[N];
                load_regs<N>(src + q * N, in_dl);
                uint32_t dd_dl = src[3 * PXW];
                SW_STAMP(T1)
                // Path starts are written into the data, not tested by the reader: the horizontal path's previous L is
                // cleared after the last pixel of a frame row (end of the step), and the record parts the row below would read across a frame
                // edge are overwritten with a path start by their producer (below, behind a wave-uniform branch that
                // touches LDS only: with register updates in it the compiler put ~20 v_mov on the common path).
                const bool st_l = lx == lx_first, st_r = lx == lx_last;
                uint32_t L0[N], L1[N], L2[N], L3[N], n0, n1, n2, n3;
                // Own record for the row below.  Its three L parts leave as soon as each is computed: a 16-byte
                // LDS store occupies the store path for ~13 cycles per wave, 15 waves x 3 of them bunched in
                // front of the barrier would leave the VALUs idle for that long; issued early they overlap with
                // the other waves' arithmetic.  (The row below reads this buffer only after the barrier.)
                // Sweep 2: the finished S of the PREVIOUS step's pixel (it waits in sp) and its lanes' first-minimum keys go to
                // the row's winner-take-all slot of that step's parity, between the paths for the same reason; the service
                // wave picks them up in the next step.
                uint32_t* dst = (r == SW_ROWS - 1) ? expo + (u & (SW_RING - 1)) * REC
                                                   : slot + ((j & 1) * (SW_ROWS + 1) + r + 1) * REC;
                uint32_t* const ws = wrow + ((j + 1) & 1) * (SW_ROWS * WSTR);
                if (SW_ABL & 12) {
#pragma unroll
                    for (int k = 0; k < N; k++) L1[k] = cb[j][k];
                    n1 = dr_dd[j & 1] ^ dr[j & 1][0];
                } else
                sweep_path<N>(cb[j], dr[j & 1], dr_dd[j & 1], P1P1, L1, n1, big0, big15);   // from (x-1, y-1): read two steps ago
                store_regs<N>(dst + 2 * PXW + q * N, L1);
                if (SECOND) store_regs<N>(ws + q * N, sp);
                SW_EARLY_FENCE
                if (SW_ABL & 8) {
#pragma unroll
                    for (int k = 0; k < N; k++) L2[k] = cb[j][k] ^ dn[k];
                    n2 = dn_dd;
                } else
                sweep_path<N>(cb[j], dn, dn_dd, P1P1, L2, n2, big0, big15);               // from (x,   y-1): read last step
                store_regs<N>(dst + PXW + q * N, L2);
                if (SECOND) ws[PXW + q] = wta_lane_key<N>(sp, q);
                SW_EARLY_FENCE
                if (SW_ABL & 8) {
#pragma unroll
                    for (int k = 0; k < N; k++) L3[k] = cb[j][k] ^ in_dl[k];
                    n3 = dd_dl;
                } else
                sweep_path<N>(cb[j], in_dl, dd_dl, P1P1, L3, n3, big0, big15);            // from (x+1, y-1): read now
                store_regs<N>(dst + q * N, L3);
                SW_EARLY_FENCE
                if (SW_ABL & 8) {
#pragma unroll
                    for (int k = 0; k < N; k++) L0[k] = cb[j][k] ^ r0[k];
                    n0 = dd0;
                } else
                sweep_path<N>(cb[j], r0, dd0, P1P1, L0, n0, big0, big15);                  // horizontal: stays in registers
                sweep_lane_min4(n0, n1, n2, n3);
                // min over d of every path: the horizontal one into every lane (this row's next step), the
                // other three only into the record (tail words 0-3: path 3, 4-7: path 2, 8-11: path 1)
                uint32_t nq;
                row16_min_1p3(n0, n3, n1, n2, n0, nq);
                // the parts of this step's record that are used later go straight into their registers
                load_regs<N>(src + PXW + q * N, dn);
                load_regs<N>(src + 2 * PXW + q * N, dr[j & 1]);
                dn_dd = src[3 * PXW + 4];
                dr_dd[j & 1] = src[3 * PXW + 8];
                SW_STAMP(T2)
                dst[3 * PXW + q] = nq + P2P2;
                if (__builtin_amdgcn_ballot_w64(st_l || st_r) != 0) {
                    // pixel 0 of a frame: nothing continues from it down-left into the previous frame's last column
                    // (part 0, tail words 0-3); last pixel: nothing continues down-right into the next frame's column 0
                    // (part 2, tail words 8-11).  L = 0, delta = P2: what a reader sees past the image edge.
                    uint32_t zz[N];
#pragma unroll
                    for (int k = 0; k < N; k++) zz[k] = 0u;
                    if (st_l) {
                        store_regs<N>(dst + q * N, zz);
                        if (q < 4) dst[3 * PXW + q] = P2P2;
                    }
                    if (st_r) {
                        store_regs<N>(dst + 2 * PXW + q * N, zz);
                        if (q >= 8 && q < 12) dst[3 * PXW + q] = P2P2;
                    }
                }
                uint32_t snew[N];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    // every term is in [0, 32767]: signed saturating adds clamp at 32767, and
                    // min(S + L, 32767) applied path by path gives the same value in any grouping
                    uint32_t s4 = pk_add_sat15(pk_add_sat15(L0[k], L1[k]), pk_add_sat15(L2[k], L3[k]));
                    if (SECOND) s4 = pk_add_sat15(s4, sb[j][k]);
                    snew[k] = s4;
                }
                const uint32_t hm = st_r ? 0u : 0xFFFFFFFFu;   // the next pixel starts a frame row: L = 0, delta = P2
#pragma unroll
                for (int k = 0; k < N; k++) r0[k] = L0[k] & hm;
                dd0 = (n0 & hm) + P2P2;
                if (SECOND) {
#pragma unroll
                    for (int k = 0; k < N; k++) sp[k] = snew[k];
                } else {
                    // Unconditional (an inactive step writes to the thread's own sink slot): behind a branch the compiler
                    // cannot count this store, assumes it away and makes the next step's wait for its C load
                    // (s_waitcnt vmcnt) cover this store as well - a write round trip exposed in every step.
                    store_regs<N>((active && !(SW_ABL & 2)) ? Sg + offq[j] : sinkp, snew);
                }
                issue(cb[j], sb[j], offq[j], actq[j], tau + j + SW_DEPTH);
                SW_STAMP(T3)
                sweep_barrier();
                SW_STAMP(T4)
                SW_PROF(pacc[0] += T1 - T0; pacc[1] += T2 - T1; pacc[2] += T3 - T2; pacc[3] += T4 - T3;)
            }
        }
        SW_PROF(if (lane == 0) for (int i = 0; i < 4; i++) a.prof[((size_t)blockIdx.x * (CW + 1) + wave) * 4 + i] = pacc[i];)
    } else {
        bool aborted = false;
#ifdef SW_SVC_PRIO
        __builtin_amdgcn_s_setprio(SW_SVC_PRIO);
#endif
        SW_PROF(unsigned long long pacc[4] = {0, 0, 0, 0};)
        for (int tau = -SW_LEAD; tau < tau_end; tau += SW_DEPTH) {
            if ((tau & 15) == 0 && (*abortp) != 0) { aborted = true; break; }   // same test, same steps as the compute waves
#pragma unroll
            for (int j = 0; j < SW_DEPTH; j++) {
                const int t = tau + j;
                SW_STAMP(T0)
                // Every vector memory operation of a step is issued by every service wave, in the same order, whatever its
                // band and whatever the pixel (a band without a neighbour and a pixel outside the stream use the dummy record
                // behind the last one; lanes past the end of a record repeat its last word): with loads or stores behind a
                // condition the compiler cannot count what is in flight and waits for ALL of it - the previous step's stores
                // included, a write round trip in every step - before it touches the record imported SW_DEPTH steps ago.
                // import: pixel t + 2 of the previous band's last row (loaded SW_DEPTH steps ago) -> ring;
                // row 0 reads pixel t' - 1 at step t'.
                if (do_import && (t & 15) == 0 && t + 23 > 0) {
                    const unsigned need = (unsigned)min(t + 23, total_px);
                    unsigned long long waited = 0, last = __builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                        __builtin_amdgcn_s_sleep(16);
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        waited += min(now - last, SW_WAIT_TICK_CAP);
                        last = now;
                        if (waited > SW_WAIT_TICKS) {
                            (*abortp) = 1;
                            if (lane == 0) __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
                {
                    unsigned long long* const w = reinterpret_cast<unsigned long long*>(imp + ((t + 2) & (SW_RING - 1)) * REC);
                    // a pixel outside the stream (the one before its first, the one after its last) and every pixel of band 0:
                    // the record of a path start (L = 0, delta = P2), which is what the rows inside a band find there
                    const bool inside = do_import && t + 2 >= 0 && t + 2 < total_px;
#pragma unroll
                    for (int i = 0; i < NQ; i++) w[li[i]] = inside ? ib[j][i] : startq[i];
                    const int p = t + 2 + SW_DEPTH;   // <= t + 6: covered by the counter value waited for (t + 23 at t % 16 == 0)
                    const unsigned long long* g = ho_in + (size_t)((p >= 0 && p < total_px) ? p : a.ho_px) * RECQ;
#pragma unroll
                    for (int i = 0; i < NQ; i++) ib[j][i] = __hip_atomic_load(g + li[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // export: the record row SW_ROWS-1 wrote in the previous step (pixel t - 1 - 2*SW_ROWS)
                {
                    const int p = t - 1 - 2 * SW_ROWS;
                    const unsigned long long* e = reinterpret_cast<const unsigned long long*>(expo + (p & (SW_RING - 1)) * REC);
                    unsigned long long* g = ho_out + (size_t)((do_export && p >= 0 && p < total_px) ? p : a.ho_px) * RECQ;
#pragma unroll
                    for (int i = 0; i < NQ; i++) __hip_atomic_store(g + li[i], e[li[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (do_export && (t & 15) == 8) {
                        // every memory operation of this wave done (stores written through), then the counter
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const int done = min(max(p + 1, 0), total_px);
                        if (lane == 0) __hip_atomic_store(flag_out, (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (SECOND && !(SW_ABL & 1)) {
                    // slot of parity t & 1: written by the row during step t - 1 (S of step t - 2); the row writes the other one now
                    const uint2 rcd = wta_from_slot<N>(wrow + (j & 1) * (SW_ROWS * WSTR), uqk, a.min_disp);
                    const bool valid = wrowok && wx >= 0 && wfi < nfr;
                    *(valid ? recg + (wrec + (uint32_t)(W1 - 1 - wx)) : reinterpret_cast<uint2*>(sinkp)) = rcd;
                    wx++;
                    const bool wrap = wx == W1;
                    wx = wrap ? 0 : wx;
                    wfi += wrap ? 1 : 0;
                    wrec += wrap ? frame_px : 0u;
                }
                SW_STAMP(T3)
                sweep_barrier();
                SW_STAMP(T4)
                SW_PROF(pacc[0] += T3 - T0; pacc[3] += T4 - T3;)
            }
        }
        SW_PROF(if (lane == 0) for (int i = 0; i < 4; i++) a.prof[((size_t)blockIdx.x * (CW + 1) + wave) * 4 + i] = pacc[i];)
        // Also after a timeout (the error word is set, the call's results are invalid anyway): the bands
        // below must not each wait out their own bound on a counter that will never move.
        (void)aborted;
        if (do_export) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flag_out, (unsigned)total_px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace svk
