// sv_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of libsv_hip.
//
// Each kernel names the reference code whose RESULT it reproduces (paths relative to
// /root/reference/modules/DisparityModule/).  None of them follows the reference's launch
// shapes: the CUDA back-end runs one 32-thread warp per scanline with the previous L in shared
// memory and a device-wide sync after every launch (sgbm_cuda/bova_sgm.cu:91-134); here a
// scanline lives in 16 lanes of a wave64 (four scanlines per wave), previous-L stays in VGPRs as
// packed u16 pairs, neighbours move with DPP row shifts and the per-pixel minimum is a 4-step DPP
// butterfly, so a step never touches LDS or a barrier.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svk {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}
// DPP controls (gfx9): within a 16-lane row.
constexpr int DPP_QUAD_XOR1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHL1 = 0x101;  // lane i <- lane i+1
constexpr int DPP_ROW_SHR1 = 0x111;  // lane i <- lane i-1
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;

// all-lanes reduction over a 16-lane row (every lane ends with the result).  `old` is the
// operation's identity: with it hipcc folds the DPP move into the ALU op (v_min_u32_dpp), one
// instruction per butterfly stage instead of two.
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
    v = min(v, dpp<DPP_QUAD_XOR1>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_QUAD_XOR2>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, v));
    // Last stage by hand: when its consumer sits in another basic block the compiler leaves this
    // stage as v_mov -1 + v_mov_dpp + v_min (three instructions).  Every lane has a mirror partner,
    // so no `old` value is needed; s_nop 1 = the two wait states between a VALU write and a DPP read.
    uint32_t o;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    return o;
}
// Four all-row minima in 10 DPP operations instead of 16: a is reduced to every lane of the row
// (the butterfly of row16_min_u32); b, c and d are reduced jointly, each stage keeping a value only
// in the quads (bank_mask = groups of 4 lanes) that go on with it: after the two cross-quad stages
// quad 0 holds b, quads 1 and 3 hold d and quad 2 holds c, and the two intra-quad stages finish
// them there.  zq = min b in lanes 0-3, min d in lanes 4-7 and 12-15, min c in lanes 8-11.
// Hand-scheduled: two other instructions (or an s_nop) between a VALU write and the DPP read of it.
__device__ __forceinline__ void row16_min_1p3(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t& a_all, uint32_t& zq) {
    uint32_t x, y, z, t;
    asm("s_nop 1\n\t"
        "v_min_u32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_min_u32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_min_u32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %3, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_min_u32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(t)
        : "v"(a), "v"(b), "v"(c), "v"(d));
    a_all = t;
    zq = z;
}
__device__ __forceinline__ uint32_t row16_or_u32(uint32_t v) {
    v |= dpp<DPP_QUAD_XOR1>(0u, v);
    v |= dpp<DPP_QUAD_XOR2>(0u, v);
    v |= dpp<DPP_ROW_HALF_MIRROR>(0u, v);
    uint32_t o;  // last stage by hand, see row16_min_u32
    asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    return o;
}
__device__ __forceinline__ uint32_t row16_add_u32(uint32_t v) {
    v += dpp<DPP_QUAD_XOR1>(0u, v);
    v += dpp<DPP_QUAD_XOR2>(0u, v);
    v += dpp<DPP_ROW_HALF_MIRROR>(0u, v);
    uint32_t o;
    asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    return o;
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {  // each half = max(a - b, 0)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// saturating packed add (v_pk_add_u16 clamp): each half = min(a + b, 65535)
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// signed saturating packed add (v_pk_add_i16 clamp): for operands in [0, 32767] each half = min(a + b, 32767),
// the S accumulation rule of aggregation.cu:135-139 in one instruction
__device__ __forceinline__ uint32_t pk_add_sat15(uint32_t a, uint32_t b) {
    typedef short ss2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b)));
}

template <int N>
__device__ __forceinline__ void load_regs(const uint32_t* __restrict__ p, uint32_t (&r)[N]) {
    if constexpr (N == 1) {
        r[0] = p[0];
    } else if constexpr (N == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        r[0] = v.x; r[1] = v.y;
    } else if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; i++) {
            uint4 v = reinterpret_cast<const uint4*>(p)[i];
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            uint2 v = reinterpret_cast<const uint2*>(p)[i];
            r[2 * i] = v.x; r[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = p[i];
    }
}
template <int N>
__device__ __forceinline__ void store_regs(uint32_t* __restrict__ p, const uint32_t (&r)[N]) {
    if constexpr (N == 1) {
        p[0] = r[0];
    } else if constexpr (N == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(r[0], r[1]);
    } else if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; i++)
            reinterpret_cast<uint4*>(p)[i] = make_uint4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; i++) reinterpret_cast<uint2*>(p)[i] = make_uint2(r[2 * i], r[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) p[i] = r[i];
    }
}

// Vector memory operations the compiler does not count (k_cost2's row loop).  hipcc places `s_waitcnt vmcnt(n)` with n the
// smallest count over every path into the wait, so a loop whose entry and back edge carry different operations in flight waits
// for more than it needs - in k_cost2 for the C store of the row before, a write round trip per image row.  Issued from
// inline assembly the operations are invisible to that pass and the waits are written by hand (vm_wait), exact: vmcnt counts
// the operations issued after the one waited for, in program order.  The destination is an in/out operand so that a slot of
// a prefetch queue keeps one register for the life of the loop, and vm_wait takes it the same way so that no use of the
// loaded value moves above the wait.  Stores: the data registers may be written two wait states after the store (s_nop 1).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vm_load_b64(u32x2& dst, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(dst) : "v"(p)); }
__device__ __forceinline__ void vm_load_b32(uint32_t& dst, const void* p) { asm volatile("global_load_dword %0, %1, off" : "+v"(dst) : "v"(p)); }
template <int CNT>
__device__ __forceinline__ void vm_wait(u32x2& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void vm_wait(uint32_t& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void vm_wait(uint32_t& a, uint32_t& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }
#ifndef SV_COST_ST_MOD
#define SV_COST_ST_MOD ""   // cache policy bits of the C store (experiments: " nt", " sc1", " sc0 sc1")
#endif
template <int N>
__device__ __forceinline__ void vm_store_regs(uint32_t* p, const uint32_t (&r)[N]) {
    static_assert(N >= 1 && N <= 4, "one store instruction");
    if constexpr (N == 1) {
        asm volatile("global_store_dword %0, %1, off" SV_COST_ST_MOD "\n\ts_nop 1" ::"v"(p), "v"(r[0]));
    } else if constexpr (N == 2) {
        const u32x2 v = {r[0], r[1]};
        asm volatile("global_store_dwordx2 %0, %1, off" SV_COST_ST_MOD "\n\ts_nop 1" ::"v"(p), "v"(v));
    } else if constexpr (N == 3) {
        const u32x3 v = {r[0], r[1], r[2]};
        asm volatile("global_store_dwordx3 %0, %1, off" SV_COST_ST_MOD "\n\ts_nop 1" ::"v"(p), "v"(v));
    } else {
        const u32x4 v = {r[0], r[1], r[2], r[3]};
        asm volatile("global_store_dwordx4 %0, %1, off" SV_COST_ST_MOD "\n\ts_nop 1" ::"v"(p), "v"(v));
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 1: prefilter + Birchfield-Tomasi interval statistics.
// Result of get_gradient (sgbm_cuda/costs.cu:116-166, clip table :286-292) followed by the
// per-pixel half-sample min/max that get_pixel_diff recomputes for every disparity
// (costs.cu:180-193).  One 8-byte record per pixel: {g, g0, g1, r, r0, r1, 0, 0} where g is the
// clipped Sobel-x value, r the raw channel with its edge columns forced to ftzero, and a0/a1 the
// min/max of {a, (a+left)/2, (a+right)/2}.
// ---------------------------------------------------------------------------------------------
// ps = pixel stride in bytes (1: gray plane, 3: one channel of an interleaved RGB image)
__device__ __forceinline__ int grad_at(const uint8_t* __restrict__ img, int W, int H, int x, int y, int cap, int ps) {
    if (x <= 0 || x >= W - 1) return cap;
    const uint8_t* r0 = img + (size_t)y * W * ps;
    const uint8_t* rm = img + (size_t)max(y - 1, 0) * W * ps;
    const uint8_t* rp = img + (size_t)min(y + 1, H - 1) * W * ps;
    const int a = (x + 1) * ps, b = (x - 1) * ps;
    int v = ((int)r0[a] - (int)r0[b]) * 2 + (int)rm[a] - (int)rm[b] + (int)rp[a] - (int)rp[b];
    return min(max(v, -cap), cap) + cap;
}
__device__ __forceinline__ int raw_at(const uint8_t* __restrict__ img, int W, int x, int y, int cap, int ps) {
    if (x <= 0 || x >= W - 1) return cap;
    return img[((size_t)y * W + x) * ps];
}

// ---------------------------------------------------------------------------------------------
// Stage 1+2: prefilter + BT statistics, then the matching cost volume C in one fused kernel.
// Result of get_gradient (costs.cu:116-166) and of get_pixel_diff x2 + get_hsum + get_cost
// (costs.cu:170-284) without materialising pixDiff or hsum and without the two memsets:
//   C[y][xl][d] = P2 + sum_{i=-r..r} sum_{k=-r..r} pix[max(y+i,0)][clamp(xl+k,0,W1-1)][d]
// which is what the reference's running sums equal when no int16 store wraps (sv_check_params),
// with its two quirks kept: the bottom SH2 rows repeat the last computed row (:279-282, or P2 with
// SV_FLAG_C_TAIL_P2) and column x = D is overwritten afterwards by k_cost_col0 (:254-267).
// k_stats_planes writes the LEFT statistics as 8-byte records {g, g0, g1, r, r0, r1} (g = clipped
// Sobel-x value, r = raw channel with its edge columns forced to ftzero, a0/a1 = min/max of
// {a, (a+left)/2, (a+right)/2}: what get_pixel_diff recomputes per disparity) and the RIGHT
// statistics as three planes of u16, reversed in x (index i = W-1-x), each in two copies (the
// second shifted by one element); an element holds a gradient statistic in its low byte and the
// matching raw statistic in its high byte ((g, r), (g0, r0), (g1, r1)).  A lane that needs the right
// statistics of x-d for 2N consecutive disparities reads N aligned dwords per plane straight from
// L1/L2; (elem(x-d), elem(x-d-1)) is the packed pair the 16-bit VALU wants after one v_and (low
// bytes) or one v_perm (high bytes).  No per-row staging of the right image in LDS, one barrier per
// image row instead of three.  (Unpacked u16 planes, one statistic each, cost twice the L1 traffic
// and 15 % more time in k_cost2.)
// Plane layout: rp[frame][pair 0..2][copy 0..1][H][WP] u16, WP even, >= W + 2.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int cost_plane_pitch(int W) { return (W + 2 + 7) & ~7; }

// cn = 1: gray images [frame][H][W].  cn = 3: interleaved RGB [frame][H][W][3]; every colour plane
// gets its own statistics (blockIdx.z = ((frame * cn + channel) * 2 + is_right)), which is what
// cv::StereoSGBM's calcPixelCostBT does with 8UC3 input (matchSGBM, StereoMatcher.cpp:392-417).
// A thread owns four consecutive "elements" of an image row; a block of T threads covers 4(T-2) of
// them plus a halo group on each side.  Element e of the left image is pixel x = e, of the right
// image pixel x = W-1-e (the reversed order its planes are stored in), so that a thread's four
// right-image statistics are one aligned 8-byte store per plane copy (2-byte stores, one pixel per
// thread, made this kernel 2.6 x slower).  Every thread computes the clipped gradient and the raw
// value of its elements (18 byte loads for four pixels), leaves them in LDS for its neighbours,
// and then the statistics of its four elements - five for the right image, whose second plane
// copy is shifted by one element.
__device__ __forceinline__ void stat_minmax(int a, int lo, int hi, bool has_lo, bool has_hi, int& a0, int& a1) {
    const int al = has_lo ? (a + lo) / 2 : a, ah = has_hi ? (a + hi) / 2 : a;
    a0 = min(min(al, ah), a);
    a1 = max(max(al, ah), a);
}
constexpr int STATS_ROWS = 8;   // image rows per block: a row's three source rows are loaded once and kept in registers
__global__ __launch_bounds__(256) void k_stats_planes(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                       uint2* __restrict__ stat_l, uint16_t* __restrict__ rp, int W, int H, int WP,
                                                       int cap, int cn) {
    __shared__ uint16_t gr_lds[2][4 * 256 + 4];
    const int T = blockDim.x, t = threadIdx.x;
    const int e0 = (int)blockIdx.x * 4 * (T - 2) - 4 + 4 * t;   // first element of this thread (halo groups: t = 0, T-1)
    const int ya = blockIdx.y * STATS_ROWS, yb = min(ya + STATS_ROWS, H);
    const int fc = blockIdx.z >> 1, frame = fc / cn, ch = fc - frame * cn;
    const bool is_right = blockIdx.z & 1;
    const uint8_t* img = (is_right ? right : left) + (size_t)frame * W * H * cn + ch;
    const bool writer = t != 0 && t != T - 1 && e0 < W;
    // the six pixels around the group (clamped reads; values outside the image are never used): rows y-1, y, y+1
    int xc[6], am[6], a0[6], ap[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int e = e0 - 1 + k, x = is_right ? W - 1 - e : e;
        xc[k] = min(max(x, 0), W - 1) * cn;
        am[k] = img[(size_t)max(ya - 1, 0) * W * cn + xc[k]];
        a0[k] = img[(size_t)ya * W * cn + xc[k]];
    }
    const int sgn = is_right ? -1 : 1;  // element e+1 is pixel x-1 in the reversed order
    for (int y = ya; y < yb; y++) {
        uint16_t* gr = gr_lds[(y - ya) & 1];
        const uint8_t* rq = img + (size_t)min(y + 1, H - 1) * W * cn;
        int col[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            ap[k] = rq[xc[k]];
            col[k] = 2 * a0[k] + am[k] + ap[k];   // Sobel-x column sum
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = e0 + k, x = is_right ? W - 1 - e : e;
            const bool edge = x <= 0 || x >= W - 1;
            const int g = edge ? cap : min(max(sgn * (col[k + 2] - col[k]), -cap), cap) + cap;
            const int r = edge ? cap : a0[k + 1];
            gr[4 * t + k] = (uint16_t)(g | (r << 8));
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { am[k] = a0[k]; a0[k] = ap[k]; }
        // one barrier per row: this row's values go to the buffer the row before last used, whose readers
        // passed the previous row's barrier
        __syncthreads();
        if (!writer) continue;
        // statistics of elements e0 .. e0+4 from gr of e0-1 .. e0+5
        int gv[7], rv[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int v = gr[4 * t - 1 + k];
            gv[k] = v & 0xFF;
            rv[k] = v >> 8;
        }
        uint32_t st[3][5];   // per element: (g, r), (g0, r0), (g1, r1) as gradient | raw << 8
        uint2 rec[4];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int e = e0 + k;
            const bool has_lo = e > 0, has_hi = e < W - 1;  // a neighbour on either side (symmetric: the order does not matter)
            int g0, g1, q0, q1;
            stat_minmax(gv[k + 1], gv[k], gv[k + 2], has_lo, has_hi, g0, g1);
            stat_minmax(rv[k + 1], rv[k], rv[k + 2], has_lo, has_hi, q0, q1);
            st[0][k] = (uint32_t)(gv[k + 1] | (rv[k + 1] << 8));
            st[1][k] = (uint32_t)(g0 | (q0 << 8));
            st[2][k] = (uint32_t)(g1 | (q1 << 8));
            if (k < 4)
                rec[k] = make_uint2((uint32_t)gv[k + 1] | ((uint32_t)g0 << 8) | ((uint32_t)g1 << 16) | ((uint32_t)rv[k + 1] << 24),
                                    (uint32_t)q0 | ((uint32_t)q1 << 8));
        }
        if (!is_right) {
            const size_t off = (size_t)fc * W * H + (size_t)y * W + e0;
            uint2* out = stat_l + off;
            if (e0 + 3 < W && (off & 1) == 0) {   // four records, 16-byte aligned
                reinterpret_cast<uint4*>(out)[0] = make_uint4(rec[0].x, rec[0].y, rec[1].x, rec[1].y);
                reinterpret_cast<uint4*>(out)[1] = make_uint4(rec[2].x, rec[2].y, rec[3].x, rec[3].y);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (e0 + k < W) out[k] = rec[k];
            }
        } else {
            uint16_t* base = rp + (size_t)fc * 6 * H * WP + (size_t)y * WP + e0;   // e0 is a multiple of 4, WP of 8: 8-byte aligned
#pragma unroll
            for (int s3 = 0; s3 < 3; s3++) {
                uint16_t* c0 = base + (size_t)(2 * s3) * H * WP;       // copy 0: element i at index i       (i <= W-1)
                uint16_t* c1 = base + (size_t)(2 * s3 + 1) * H * WP;   // copy 1: element i at index i - 1   (indices 0 .. W-2)
                if (e0 + 3 < W) *reinterpret_cast<uint2*>(c0) = make_uint2(st[s3][0] | (st[s3][1] << 16), st[s3][2] | (st[s3][3] << 16));
                else
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (e0 + k < W) c0[k] = (uint16_t)st[s3][k];
                if (e0 + 4 < W) *reinterpret_cast<uint2*>(c1) = make_uint2(st[s3][1] | (st[s3][2] << 16), st[s3][3] | (st[s3][4] << 16));
                else
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (e0 + k + 1 < W) c1[k] = (uint16_t)st[s3][k + 1];
            }
        }
    }
}

// BT cost of one channel for two disparities: distance of u to [v0,v1] and of v to [u0,u1], the
// smaller of the two (costs.cu:180-196).  v0 <= v1, so at most one of the two clamped differences
// is non-zero and an OR joins them.
__device__ __forceinline__ uint32_t bt_pk2(uint32_t u, uint32_t u0, uint32_t u1, uint32_t v, uint32_t v0, uint32_t v1) {
    const uint32_t c0 = pk_sub_sat(u, v1) | pk_sub_sat(v0, u);
    const uint32_t c1 = pk_sub_sat(v, u1) | pk_sub_sat(u0, v);
    return pk_min(c0, c1);
}

// Workgroup barrier that orders LDS traffic only (global loads stay in flight across it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A workgroup of NT threads owns NT/16 tile columns (TX = NT/16 - 2r output columns + the halo of
// the horizontal window) x all D and marches down a band of rows.  Thread (j = tid/16, q = tid%16):
// BT phase: the costs of tile column j for its 2N disparities -> LDS; window phase (j < TX): the
// 2r+1-tap horizontal sum for output column j from LDS, the last 2r+2 sums in a private LDS ring,
// the vertical running sum in registers, C out with the 16-byte stores k_agg / k_sweep read back.
#ifndef SV_COST_KATTR
#define SV_COST_KATTR   // experiments: extra attributes of k_cost2, e.g. __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef SV_COST_ST_SPREAD
#define SV_COST_ST_SPREAD 1   // the odd waves of a workgroup store their finished row of C at the top of a trip, the even ones at its end: the
                              // workgroup's 15 KB do not reach the L1 in one burst (20.9 -> 20.6 ms per 1024 C2 frames; same wait counts,
                              // the trip's three operations are the same cycle S, L, V entered at another point)
#endif
#ifndef SV_COST_TAPS_EARLY
#define SV_COST_TAPS_EARLY 15   // window taps requested before the BT costs of the next row (the rest after them: fewer registers live across the costs)
#endif
#ifndef SV_COST_DEPTH
#define SV_COST_DEPTH 2   // image rows the gray cost kernel's loads run ahead of their use (k_cost2, STAGE path)
#endif
#ifndef SV_COST_ABL
#define SV_COST_ABL 0   // diagnostic builds (wrong results): 1 C to the sink, 4 one window tap, 8 no BT arithmetic, 16 no ring
                        // (-DSV_COST_DRAIN: every wait of the queue a full drain, right results: tells a wrong count from a wrong index)
#endif
constexpr int COST_SINK_SLOTS = 256;   // sink slots of NT * N dwords each (workgroup id mod 256)
// Gray cost (CN = 1): the right-image statistics a tile row needs - NCOL + 32 N - 1 consecutive elements of each of the
// three reversed planes, 1.2 KB at D = 128 - are staged ONCE per image row in LDS, already unpacked into the u16 pairs the
// packed VALU wants: [3 statistics a, a0, a1][4 copies][SPANW] pairs of dwords {gradient pair, raw pair}, entry m of copy c =
// (element 2m + c, element 2m + c + 1) counted from the tile row's first element rounded down to a multiple of 4; a lane
// whose first element is i0 reads copy i0 mod 4 from an even entry on, i.e. with 16-byte aligned ds_read_b128.  Before, every lane loaded its 3 N packed dwords through
// L1 (49 KB per tile row for 1.2 KB of distinct data) and spent 6 N instructions unpacking them, per (column, disparity)
// instead of per element.  Double-buffered by the row's parity.
template <int N, int NT>
__host__ __device__ constexpr int cost_stage_spanw() {
    // entries per copy; = 2 mod 4, so that a copy is 4 mod 8 dwords long and neighbouring columns (which read neighbouring copies)
    // fall on the other half of every 8 banks in a ds_read_b128 lane group
    return ((NT / 16 + 32 * N + 8) / 2) % 4 == 0 ? (NT / 16 + 32 * N + 8) / 2 + 2 : (NT / 16 + 32 * N + 8) / 2 + ((NT / 16 + 32 * N + 8) / 2) % 2;
}
template <int N, int NT>
__host__ __device__ constexpr int cost_stage_items() { return (6 * (cost_stage_spanw<N, NT>() + 1) + NT - 1) / NT; }   // global dwords per thread and row
// ... where it pays: up to D = 128 (measured: 640x480 D=128 24.4 -> 23.8 ms per 1024 frames, 320x240 D=64 5.05 -> 4.65; at D = 256, two
// waves per SIMD and 222 VGPRs, 38.3 -> 43.4 ms per 98 frames of 1920x1080, so those keep the direct loads)
template <int N, int CN>
__host__ __device__ constexpr bool cost_stage() { return CN == 1 && N <= 4; }
template <int N, int NT, int CN>
__host__ __device__ constexpr size_t cost_lds_bytes(int radius) {
    const int NCOL = NT / 16, TX = NCOL - 2 * radius;
    return ((size_t)(2 * NCOL + (2 * radius + 1) * TX) * (16 * N) + (cost_stage<N, CN>() ? (size_t)2 * 24 * cost_stage_spanw<N, NT>() : 0)) * 4;
}
template <int N, int NT, int CN>
__global__ __launch_bounds__(NT) SV_COST_KATTR void k_cost2(const uint2* __restrict__ stat_l, const uint16_t* __restrict__ rp,
                                               uint32_t* __restrict__ C, int W, int H, int WP, int radius, int yband,
                                               uint32_t P2P2, int tail_p2, int ntx, int nbands, int nframes, int col0, int W1,
                                               int min_disp, int cs, int dpart0, int nparts, uint32_t* __restrict__ sink) {
    constexpr int cn = CN;  // colour planes summed into the cost (1 = gray, 3 = RGB)
    constexpr bool STAGE = cost_stage<N, CN>();   // right statistics through the LDS stage (cost_stage_spanw); the colour cost and D > 128 keep the direct loads
    extern __shared__ __align__(16) uint32_t c2_lds[];
    constexpr int D2 = 16 * N, NCOL = NT / 16;
    constexpr int SPANW = cost_stage_spanw<N, NT>(), ITEMS = cost_stage_items<N, NT>(), RSBUF = 24 * SPANW;
    // col0 = first computable column, W1 = their number, min_disp = minDisparity (EXT, cv::StereoSGBM): cost index d
    // pairs left x with right x - d - min_disp; with min_disp = 0 (the CUDA back-end) col0 = D, W1 = W - D
    const int tid = threadIdx.x, j = tid >> 4, q = tid & 15;
    const int TX = NCOL - 2 * radius, ringn = 2 * radius + 1;
    uint32_t* const pix = c2_lds;                      // [2][NCOL][D2]
    uint32_t* const ring = pix + 2 * NCOL * D2;        // [ringn][TX][D2]
    uint32_t* const rs = ring + ringn * TX * D2;       // STAGE: [2][3][4][SPANW] x {gradient pair, raw pair}
    // 1-D grid, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so id % 8
    // picks the frame within a group of 8 and all column tiles of a frame (whose right-image reads
    // overlap by D + 2r columns) share one XCD's L2.
    // Fewer than 8 frames (nframes < 0 on entry: plain order): a frame's tiles spread over all XCDs.
    // Disparity parts: a workgroup costs 32 N disparities, the volume may hold more (cs = dwords of a pixel in C): part p of
    // the launch covers disparities dpart0 + 32 N p onwards and writes that slice of every pixel's line.  D = 256 runs as
    // two parts of N = 4 on the 64-column tiles and the LDS stage of the D <= 128 kernel (its own N = 8 form has two waves
    // per SIMD and 32-column tiles); the parts of a tile are neighbours in the grid (same frame, same XCD).
    const int per_frame = ntx * nbands * nparts;
    const int wg = blockIdx.x;
    const bool by_xcd = nframes > 0;
    const int fr = by_xcd ? (wg / 8 / per_frame) * 8 + (wg & 7) : wg / per_frame;
    const int tbp = by_xcd ? (wg / 8) % per_frame : wg % per_frame;
    const int tb = tbp / nparts, doff = dpart0 + (tbp % nparts) * (32 * N);
    min_disp += doff;   // from here on: the shift between a cost index of this part and the right image's column
    if (fr >= (by_xcd ? nframes : -nframes)) return;
    const int xl0 = (tb % ntx) * TX;
    const int y0 = (tb / ntx) * yband, y1 = min(y0 + yband, H);
    stat_l += (size_t)fr * cn * W * H;
    const uint32_t* rp32 = reinterpret_cast<const uint32_t*>(rp + (size_t)fr * cn * 6 * H * WP);
    C += (size_t)fr * H * W1 * cs + doff / 2;
    // BT item of this thread: tile column j (clamped into the computable band), disparities 2N*q ..
    const int xj = min(max(xl0 - radius + j, 0), W1 - 1) + col0;
    const int e = (W - 1 - xj) + min_disp + 2 * N * q;  // reversed index of x - d - min_disp for the first disparity
    const int cpy = e & 1;
    const size_t plane = (size_t)H * WP / 2;           // dwords per plane copy
    const uint32_t* rbase = rp32 + (size_t)cpy * plane + (size_t)((e - cpy) >> 1);
    const int xl = xl0 + j;
    const bool colok = j < TX && xl < W1;
    uint32_t* const mypix = pix + j * D2 + q * N;
    uint32_t* const myring = ring + j * D2 + q * N;

    // ---- the stage.  First element of the tile row: that of its last column (the planes are reversed in x) at d = 0,
    // rounded down to a multiple of 4.  Item it (0 .. 6 (SPANW + 1)) = (plane, global copy gc, dword mm): global dword
    // e_lo4 / 2 + mm of copy gc = elements (e_lo4 + 2 mm + gc, + 1) = dword mm of stage copy gc and dword mm - 1 of copy gc + 2.
    const int xj_max = min(max(xl0 - radius + NCOL - 1, 0), W1 - 1) + col0;
    const int e_lo4 = ((W - 1 - xj_max) + min_disp) & ~3;
    const int i0 = e - e_lo4;                            // this lane's first element within the stage
    const uint32_t* const myrs = rs + 2 * ((i0 & 3) * SPANW + ((i0 - (i0 & 3)) >> 1));   // + statistic * 8 * SPANW + parity * RSBUF
    uint32_t sv[ITEMS];        // staged dwords in flight (one image row ahead of the stage's contents)
    size_t sv_src[ITEMS];      // their dword offset in row 0 of the planes (plane and copy included, clamped into the row's pitch)
    int sv_dst[ITEMS];         // stage entry, copy gc: (plane * 4 + gc) * SPANW + mm
    int sv_mm[ITEMS];          // mm, or -1: this thread has no such item
    if (STAGE) {
#pragma unroll
        for (int t = 0; t < ITEMS; t++) {
            const int it = min(tid + t * NT, 6 * (SPANW + 1) - 1);
            const int plc = it / (SPANW + 1), mm = it - plc * (SPANW + 1);   // plc = 2 * plane + global copy
            sv_src[t] = (size_t)plc * plane + (size_t)min(e_lo4 / 2 + mm, WP / 2 - 1);   // (past the row's pitch: a value nobody reads)
            sv_dst[t] = ((plc >> 1) * 4 + (plc & 1)) * SPANW + mm;
            sv_mm[t] = tid + t * NT < 6 * (SPANW + 1) ? mm : -1;
            sv[t] = 0;
        }
    }
    // every thread loads (those without an item repeat the last one): no vector memory operation behind a condition
    auto stage_load = [&](int ry) {
        const uint32_t* rowp = rp32 + (size_t)min(ry, H - 1) * (WP / 2);
#pragma unroll
        for (int t = 0; t < ITEMS; t++) sv[t] = rowp[sv_src[t]];
    };
    auto stage_write = [&](int par, const uint32_t (&dw)[ITEMS]) {
        uint32_t* const dstb = rs + par * RSBUF;
#pragma unroll
        for (int t = 0; t < ITEMS; t++) {
            const uint32_t g = dw[t] & 0x00FF00FFu, r = __builtin_amdgcn_perm(0u, dw[t], 0x0C030C01u);
            const int d0 = sv_dst[t], mm = sv_mm[t];
            if (mm >= 0 && mm < SPANW) *reinterpret_cast<uint2*>(dstb + 2 * d0) = make_uint2(g, r);                    // copy gc, entry mm
            if (mm >= 1) *reinterpret_cast<uint2*>(dstb + 2 * (d0 + 2 * SPANW - 1)) = make_uint2(g, r);                 // copy gc + 2, entry mm - 1
        }
    };

    // the lane's right statistics of the NEXT row to be costed, as u16 pairs (g, g0, g1, r, r0, r1), read from the stage one
    // trip ahead so that the costs start from registers and the reads' latency lies under the window sums
    uint32_t st[6][N];
#pragma unroll
    for (int s6 = 0; s6 < 6; s6++)
#pragma unroll
        for (int k = 0; k < N; k++) st[s6][k] = 0;
    auto stage_read = [&](int par) {
        const uint32_t* src = myrs + par * RSBUF;
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) {
            // 16-byte aligned by construction (copy = first element mod 4, so the first entry is even): two entries per read
#pragma unroll
            for (int k = 0; k + 1 < N; k += 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + s3 * 8 * SPANW + 2 * k);
                st[s3][k] = v.x; st[3 + s3][k] = v.y; st[s3][k + 1] = v.z; st[3 + s3][k + 1] = v.w;
            }
            if (N & 1) {
                const uint2 v = *reinterpret_cast<const uint2*>(src + s3 * 8 * SPANW + 2 * (N - 1));
                st[s3][N - 1] = v.x; st[3 + s3][N - 1] = v.y;
            }
        }
    };
    uint2 lrec;
    uint32_t rq[3][N];  // !STAGE: right statistics as loaded: per u16 the gradient statistic in the low byte, the raw one in the high byte
    auto prefetch = [&](int ry, int ch) {
        ry = min(ry, H - 1);
        lrec = stat_l[(size_t)ch * W * H + (size_t)ry * W + xj];
        if (!STAGE) {
#pragma unroll
            for (int s3 = 0; s3 < 3; s3++)
                load_regs<N>(rbase + (size_t)(6 * ch + 2 * s3) * plane + (size_t)ry * (WP / 2), rq[s3]);
        }
    };
    // BT costs of image row ry, summed over the cn colour planes (operands of plane 0 in lrec / rq,
    // prefetched) -> pix[ry & 1]; while a plane is computed the operands of the next plane, or of
    // plane 0 of the next row, are on their way.
    auto bt_store = [&](int ry, auto&& unpacked) {   // unpacked(): called once the left record is in its six registers
        uint32_t pv[N];
#pragma unroll
        for (int k = 0; k < N; k++) pv[k] = 0;
#pragma unroll 1  // one plane's operands in registers at a time (cn = 1: no loop at all)
        for (int ch = 0; ch < cn; ch++) {
            const uint32_t ug = (lrec.x & 0xFFu) * 0x10001u, ug0 = ((lrec.x >> 8) & 0xFFu) * 0x10001u;
            const uint32_t ug1 = ((lrec.x >> 16) & 0xFFu) * 0x10001u, ur = (lrec.x >> 24) * 0x10001u;
            const uint32_t ur0 = (lrec.y & 0xFFu) * 0x10001u, ur1 = ((lrec.y >> 8) & 0xFFu) * 0x10001u;
            unpacked();
#pragma unroll
            for (int k = 0; k < N; k++) {  // raw channel >> 2 (diff_scale, bova_sgm.cu:97); halves cannot carry
                uint32_t vg, vg0, vg1, vr, vr0, vr1;
                if (STAGE) {
                    vg = st[0][k]; vg0 = st[1][k]; vg1 = st[2][k]; vr = st[3][k]; vr0 = st[4][k]; vr1 = st[5][k];
                } else {
                    // unpack: one v_and / one v_perm per statistic and register (half the L1 traffic of unpacked planes)
                    vg = rq[0][k] & 0x00FF00FFu; vg0 = rq[1][k] & 0x00FF00FFu; vg1 = rq[2][k] & 0x00FF00FFu;
                    vr = __builtin_amdgcn_perm(0u, rq[0][k], 0x0C030C01u); vr0 = __builtin_amdgcn_perm(0u, rq[1][k], 0x0C030C01u);
                    vr1 = __builtin_amdgcn_perm(0u, rq[2][k], 0x0C030C01u);
                }
                if (SV_COST_ABL & 8) pv[k] += (ug ^ vg) + (ug0 ^ vg0) + (ug1 ^ vg1) + (ur ^ vr) + (ur0 ^ vr0) + (ur1 ^ vr1);
                else
                pv[k] += bt_pk2(ug, ug0, ug1, vg, vg0, vg1) + __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, bt_pk2(ur, ur0, ur1, vr, vr0, vr1)) >> 2);
            }
            if (ch + 1 < cn) prefetch(ry, ch + 1);
            else if (!STAGE) prefetch(ry + 1, 0);   // (STAGE: the row loop keeps its own queue of left records)
        }
        store_regs<N>(mypix + (ry & 1) * (NCOL * D2), pv);
    };

    const int ylast = H - 1 - radius;
    const int ys = min(y0, ylast);
    uint32_t* Cout = C + (size_t)min(xl, W1 - 1) * cs + q * N;
    uint32_t* const sinkp = sink + ((size_t)(blockIdx.x & (COST_SINK_SLOTS - 1)) * NT + tid) * N;   // shared by workgroups: never read
    const size_t rowN = (size_t)W1 * cs;
    // The rows of the band as a software pipeline, one LDS-only barrier per image row: after the
    // barrier the window taps of row ry are requested from pix[ry & 1] (all of them in flight: the
    // tap count is a compile-time constant of the body), the BT costs of row ry + 1 are computed
    // into the other half while they arrive, then the taps are summed, the row sum goes into the
    // thread's private ring and the vertical running sum moves one row down.
    // The stage runs further ahead: during the trip of row ry the BT costs of row ry + 1 start from registers filled in the
    // trip before, the statistics of row ry + 2 are read from stage[ry & 1] into those registers afterwards, the dwords of
    // row ry + 3 (loaded in the trip before) are unpacked into stage[(ry + 1) & 1], whose readers passed the barrier, and the
    // dwords of row ry + 4 are requested.
    auto run = [&](auto taps_tag) {
        constexpr int TAPS = decltype(taps_tag)::value;
        const int r_begin = max(ys - radius, 0), r_init_end = ys + radius;
        const int r_end = max(min(y1 - 1, ylast), ys) + radius;
        uint32_t acc[N], outp[N];
        uint32_t* outaddr = sinkp;
#pragma unroll
        for (int k = 0; k < N; k++) acc[k] = outp[k] = 0;
        // window taps of row ry, requested from pix[ry & 1] right after the barrier
        auto taps_load = [&](int ry, uint32_t (&v)[TAPS][N], int t0 = 0, int t1 = 99) {
            const uint32_t* pw = mypix + (ry & 1) * (NCOL * D2);
            if (j < TX) {
#pragma unroll
                for (int t = 0; t < ((SV_COST_ABL & 4) ? 1 : TAPS); t++)
                    if (t >= t0 && t < t1) load_regs<N>(pw + t * 16 * N, v[t]);
            }
        };
        // window phase of row ry: horizontal sum of the taps, ring, vertical running sum; leaves the finished row of C in
        // outp / outaddr (stored in the NEXT trip)
        auto window = [&](int ry, const uint32_t (&v)[TAPS][N]) {
            if (j < TX) {
                uint32_t hs[N];
#pragma unroll
                for (int k = 0; k < N; k++) hs[k] = 0;
#pragma unroll
                for (int t = 0; t < ((SV_COST_ABL & 4) ? 1 : TAPS); t++)
#pragma unroll
                    for (int k = 0; k < N; k++) hs[k] += v[t][k];  // <= (2 cap + 63) * 15 <= 4755 per half: no carry
                // the ring holds the last 2r + 1 row sums: the oldest one leaves (it is read first) where the new one goes
                uint32_t* const slot = myring + (ry % ringn) * (TX * D2);
                if (ry <= r_init_end) {
                    const uint32_t mult = ry == 0 ? (uint32_t)(radius - ys + 1) : 1u;  // row 0 stands for every row above it
#pragma unroll
                    for (int k = 0; k < N; k++) acc[k] += hs[k] * mult;
                } else if (SV_COST_ABL & 16) {
#pragma unroll
                    for (int k = 0; k < N; k++) acc[k] = acc[k] + hs[k];
                } else {
                    uint32_t old[N];
                    load_regs<N>(myring + (max(ry - 2 * radius - 1, 0) % ringn) * (TX * D2), old);
                    asm volatile("" ::: "memory");   // the store below stays behind this load (same slot when ry > 2r)
#pragma unroll
                    for (int k = 0; k < N; k++) acc[k] = acc[k] + hs[k] - old[k];
                }
                if (!(SV_COST_ABL & 16)) store_regs<N>(slot, hs);
            }
            const int y = ry - radius;
#pragma unroll
            for (int k = 0; k < N; k++) outp[k] = acc[k] + P2P2;
            outaddr = (ry >= r_init_end && y >= y0 && colok && !(SV_COST_ABL & 1)) ? Cout + (size_t)y * rowN : sinkp;
        };
        if constexpr (STAGE) {
            // Three vector memory streams per trip - the left record of a row to come, the stage's dwords of a row to come,
            // the finished row of C - issued from inline assembly in that cyclic order (odd waves enter the cycle at the store:
            // SV_COST_ST_SPREAD), loads DEPTH trips ahead of their use, and
            // waited for by hand: the wait for a load lets the K * DEPTH - 1 younger operations stay in flight, so a store
            // has DEPTH trips to be acknowledged before a wait covers it.  (Left to the compiler the waits were vmcnt(1) where
            // vmcnt(2) would do - the loop's entry and its back edge carried the operations in different orders - and every
            // row waited out the store of the row before: under the write stream's load that, not arithmetic, set the pace.)
            constexpr int DEPTH = SV_COST_DEPTH, K = ITEMS + 2;
            static_assert(ITEMS >= 1 && ITEMS <= 2 && DEPTH >= 1 && DEPTH <= 4 && K * DEPTH - 1 <= 63, "queue shape");
            stage_load(r_begin);
            stage_write(r_begin & 1, sv);
            stage_load(r_begin + 1);
            stage_write((r_begin + 1) & 1, sv);
            stage_load(r_begin + 2);
            __syncthreads();
            stage_read(r_begin & 1);
            prefetch(r_begin, 0);
            bt_store(r_begin, [] {});
            stage_read((r_begin + 1) & 1);
            __syncthreads();                 // every lane has row r_begin's statistics in registers: its buffer takes row r_begin + 2
            stage_write(r_begin & 1, sv);
            u32x2 lq[DEPTH];                 // left records of rows ry + 1 .. ry + DEPTH (slot = trip mod DEPTH)
            uint32_t sq[DEPTH][ITEMS];       // stage dwords of rows ry + 3 .. ry + 2 + DEPTH
            const uint2* const lbase = stat_l + xj;
            auto q_lrec = [&](u32x2& d, int row) { vm_load_b64(d, lbase + (size_t)min(row, H - 1) * W); };
            auto q_stage = [&](uint32_t (&d)[ITEMS], int row) {
                const uint32_t* rowp = rp32 + (size_t)min(row, H - 1) * (WP / 2);
#pragma unroll
                for (int t = 0; t < ITEMS; t++) vm_load_b32(d[t], rowp + sv_src[t]);
            };
#pragma unroll
            for (int p = 0; p < DEPTH; p++) {   // the loop is entered with what its back edge carries in flight
                lq[p] = u32x2{0u, 0u};
#pragma unroll
                for (int t = 0; t < ITEMS; t++) sq[p][t] = 0;
                q_lrec(lq[p], r_begin + 1 + p);
                q_stage(sq[p], r_begin + 3 + p);
                vm_store_regs<N>(sinkp, outp);
            }
            // Trip of row ry: after the barrier pix[ry & 1] is complete (its previous contents, row ry - 2, were consumed
            // before the barrier of the trip before) and its taps are requested; the BT costs of row ry + 1 are computed from
            // registers (left record from the queue, right statistics read from the stage in the trip before) while they
            // arrive; the stage's buffer of row ry + 2 goes to registers, the dwords of row ry + 3 are unpacked into the
            // buffer that row ry + 1 left; then the window sums of row ry.
#if SV_COST_ST_SPREAD
            const bool st_early = ((tid >> 6) & 1) != 0;
#endif
            auto trip = [&](int ry, auto ptag) {
                constexpr int p = decltype(ptag)::value;
                lds_barrier();   // LDS only: __syncthreads() would drain the queues
#if SV_COST_ST_SPREAD
                if (st_early) vm_store_regs<N>(outaddr, outp);   // (SV_COST_ST_SPREAD)
#endif
                uint32_t v[TAPS][N];
                taps_load(ry, v, 0, SV_COST_TAPS_EARLY);
#ifdef SV_COST_DRAIN
                vm_wait<0>(lq[p]);
#else
                vm_wait<K * DEPTH - 1>(lq[p]);
#endif
                lrec = make_uint2(lq[p].x, lq[p].y);
                bt_store(ry + 1, [&] { q_lrec(lq[p], ry + 1 + DEPTH); });   // (the slot is free once the record is unpacked)
                taps_load(ry, v, SV_COST_TAPS_EARLY, TAPS);
                stage_read(ry & 1);
#ifdef SV_COST_DRAIN
                if constexpr (ITEMS == 1) vm_wait<0>(sq[p][0]);
                else vm_wait<0>(sq[p][0], sq[p][1]);
#else
                if constexpr (ITEMS == 1) vm_wait<K * DEPTH - ITEMS>(sq[p][0]);
                else vm_wait<K * DEPTH - ITEMS>(sq[p][0], sq[p][1]);
#endif
                stage_write((ry + 1) & 1, sq[p]);
                q_stage(sq[p], ry + 3 + DEPTH);
#if SV_COST_ST_SPREAD
                if (!st_early)
#endif
                vm_store_regs<N>(outaddr, outp);   // the row finished in the trip before
                if (ry <= r_end) window(ry, v);    // (trips past the band's last row: see below)
            };
            // Whole rounds of DEPTH trips, one exit at the bottom: a queue slot is then one value that lives in one register
            // around the loop.  (With an exit after every trip the compiler kept some slots in two registers and copied between
            // them - copies of a register whose load is still in flight.  tests/test_cost_queue_isa.py checks the listing for
            // any such copy.)  The up to DEPTH - 1 trips past the last row cost rows clamped to the image, store the last row
            // of C once more and leave the sums alone.
            const int rounds = (r_end - r_begin + DEPTH) / DEPTH;
            for (int it = 0, ry = r_begin; it < rounds; it++, ry += DEPTH) {
                trip(ry, std::integral_constant<int, 0>{});
                if constexpr (DEPTH > 1) trip(ry + 1, std::integral_constant<int, 1>{});
                if constexpr (DEPTH > 2) trip(ry + 2, std::integral_constant<int, 2>{});
                if constexpr (DEPTH > 3) trip(ry + 3, std::integral_constant<int, 3>{});
            }
#pragma unroll
            for (int p = 0; p < DEPTH; p++) {   // loads still in flight land in registers that stay theirs until here
                vm_wait<0>(lq[p]);
#pragma unroll
                for (int t = 0; t < ITEMS; t++) vm_wait<0>(sq[p][t]);
            }
        } else {
            prefetch(r_begin, 0);
            bt_store(r_begin, [] {});
            for (int ry = r_begin; ry <= r_end; ry++) {
                // pix[ry & 1] complete; its previous contents (row ry-2) were consumed before the last
                // barrier.  LDS only: __syncthreads() would also wait for the prefetch in flight (vmcnt(0)).
                lds_barrier();
                // The row finished in the previous trip is stored here, at the top of the next one, and unconditionally (a thread
                // with nothing to write - warm-up rows, halo columns - stores to its sink slot).  That keeps the vector memory
                // operations in flight at the loop's entry and at its back edge the same (the prefetch only) and none of them
                // behind a branch, so the compiler counts them exactly: the wait for the prefetched operands below is vmcnt(1)
                // and leaves this store in flight.  With the store at the end of the trip behind its condition the wait was
                // vmcnt(0) and every row paid a write round trip (27.1 -> 24.0 ms per 1024 C2 frames).
                store_regs<N>(outaddr, outp);
                uint32_t v[TAPS][N];
                taps_load(ry, v);
                bt_store(ry + 1, [] {});   // also after the last row (row clamped, result unused): no vector memory operation of the loop sits behind a branch
                window(ry, v);
            }
        }
        store_regs<N>(outaddr, outp);
        // the bottom `radius` rows repeat the last computed row (costs.cu:279-282), or hold P2
        for (int y = max(y0, ylast + 1); y < y1; y++) {
            uint32_t out[N];
#pragma unroll
            for (int k = 0; k < N; k++) out[k] = tail_p2 ? P2P2 : acc[k] + P2P2;
            if (colok) store_regs<N>(Cout + (size_t)y * rowN, out);
        }
    };
    switch (radius) {
        case 0: run(std::integral_constant<int, 1>{}); break;
        case 1: run(std::integral_constant<int, 3>{}); break;
        case 2: run(std::integral_constant<int, 5>{}); break;
        case 3: run(std::integral_constant<int, 7>{}); break;
        case 4: run(std::integral_constant<int, 9>{}); break;
        case 5: run(std::integral_constant<int, 11>{}); break;
        case 6: run(std::integral_constant<int, 13>{}); break;
        default: run(std::integral_constant<int, 15>{}); break;
    }
}

// Column x = D of C (local column 0): "copy opencv" quirk, costs.cu:254-267 — every row repeats row
// 0 (REPO) or holds the initial P2 (MODE_HH, SV_FLAG_C_COL0_P2).  Runs after k_cost.
__global__ __launch_bounds__(256) void k_cost_col0(uint32_t* __restrict__ C, int W1, int H, int D2, uint32_t P2P2, int col0_p2) {
    const int dq = threadIdx.x, y = blockIdx.x + 1;
    if (dq >= D2 || y >= H) return;
    C += (size_t)blockIdx.y * H * W1 * D2;
    C[(size_t)y * W1 * D2 + dq] = col0_p2 ? P2P2 : C[dq];
}

// ---------------------------------------------------------------------------------------------
// Winner-take-all + uniqueness + sub-pixel on one pixel's S held in the k_agg register layout
// (16 lanes, lane q = disparities [q*2N, (q+1)*2N)).  Result of get_disparity
// (aggregation.cu:629-689): first minimum wins (packed key S<<16|d under a DPP min), a pixel is
// rejected if any d outside best+-1 has S*(100-uniq) < minS*100, sub-pixel by the parabola through
// S[best-1], S[best], S[best+1] with C integer division.  Returns the 8-byte WTA record
// {disp16, raw d (-1 = none), minS, 0}; every lane of the group gets the same record.
// ---------------------------------------------------------------------------------------------
// The uniqueness test "S[d]*(100-uniq) < minS*100" is done as S[d] < T with
// T = ceil(minS*100 / (100-uniq)), the division by the (uniform) constant through a magic multiply.
struct UniqK {
    uint32_t k, add, magic;  // T = umulhi(minS * k + add, magic), branch-free for every ratio
};
__device__ __forceinline__ UniqK make_uniq(int uniq) {
    const int mul = 100 - uniq;  // 100 - uniquenessRatio
    UniqK u;
    if (mul > 1) {  // ceil(minS*100 / mul): magic = floor(2^32 / mul) + 1 is exact for numerators < 2^22
        u.k = 100u; u.add = (uint32_t)mul - 1u; u.magic = 0xFFFFFFFFu / (uint32_t)mul + 1u;
    } else if (mul == 1) {  // minS*100 itself (the magic of 1 would be 2^32): floor((200 minS + 1) / 2)
        u.k = 200u; u.add = 1u; u.magic = 0x80000000u;
    } else {        // ratio 100: every S counts as soon as minS > 0: T = (minS << 16) - 1 >= 0xFFFF, or 0
        u.k = 0x10000u; u.add = 0u; u.magic = 0xFFFFFFFFu;
    }
    return u;
}

// The reductions of one pixel: every lane of the 16-lane group ends up with the same three words.
struct WtaPart {
    uint32_t key;  // minS << 16 | min(hits, 255) << 8 | best d; hits = number of disparities with S < T
    uint32_t nb;   // S[best-1] << 16 | S[best+1] (0 where there is no such disparity)
    uint32_t m2;   // wta_finish<D, true> only: the smallest S outside best-1 .. best+1 (hits field of key unused)
};
typedef uint16_t __attribute__((may_alias)) u16_alias;   // halfword view of LDS words (type-based alias analysis off)
__device__ __forceinline__ uint32_t wta_threshold(uint32_t minS, UniqK uq) {
    return __umulhi(minS * uq.k + uq.add, uq.magic);
}

// all-lanes row reductions written with the DPP builtins only (no inline asm): inside one basic block the compiler
// folds every stage into one v_*_dpp and is free to fill the two wait states a DPP read needs with other work
__device__ __forceinline__ uint32_t row16_min_u32_s(uint32_t v) {
    v = min(v, dpp<DPP_QUAD_XOR1>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_QUAD_XOR2>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_ROW_HALF_MIRROR>(0xFFFFFFFFu, v));
    return min(v, dpp<DPP_ROW_MIRROR>(0xFFFFFFFFu, v));
}
__device__ __forceinline__ uint32_t row16_add_u32_s(uint32_t v) {
    v += dpp<DPP_QUAD_XOR1>(0u, v);
    v += dpp<DPP_QUAD_XOR2>(0u, v);
    v += dpp<DPP_ROW_HALF_MIRROR>(0u, v);
    return v + dpp<DPP_ROW_MIRROR>(0u, v);
}
__device__ __forceinline__ uint32_t row16_or_u32_s(uint32_t v) {
    v |= dpp<DPP_QUAD_XOR1>(0u, v);
    v |= dpp<DPP_QUAD_XOR2>(0u, v);
    v |= dpp<DPP_ROW_HALF_MIRROR>(0u, v);
    return v | dpp<DPP_ROW_MIRROR>(0u, v);
}

// The three reductions of wta_reduce as separate stages (k_sweep spreads them over a step, next to the path
// arithmetic of the following pixel, so that their dependent DPP chains do not sit idle on wait states).
// A: first minimum: packed key (S << 16 | d) under a 32-bit min.  Inside the lane the key carries the position within
// the lane (an inline constant: no register per position); the lane's base is added to the lane's minimum.
template <int N, bool SCHED>
__device__ __forceinline__ uint32_t wta_stage_key(const uint32_t (&s)[N], int q) {
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < N; k++) {
        key = min(key, (s[k] << 16) | (uint32_t)(2 * k));
        key = min(key, (s[k] & 0xFFFF0000u) | (uint32_t)(2 * k + 1));
    }
    key += (uint32_t)(q * 2 * N);   // 2N - 1 + dbase < 256: no carry into S
    return SCHED ? row16_min_u32_s(key) : row16_min_u32(key);
}
// The lane-local part of A alone (k_sweep: the 16 lanes of a row leave their keys in LDS and one lane of the service wave
// takes the minimum of the 16).
template <int N>
__device__ __forceinline__ uint32_t wta_lane_key(const uint32_t (&s)[N], int q) {
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < N; k++) {
        key = min(key, (s[k] << 16) | (uint32_t)(2 * k));
        key = min(key, (s[k] & 0xFFFF0000u) | (uint32_t)(2 * k + 1));
    }
    return key + (uint32_t)(q * 2 * N);
}
// B: uniqueness: no S outside best-1..best+1 may be below T.  Every S lies in [0, 32767], so with T
// clamped to 32768 "S < T" is the sign of the 16-bit difference: count the hits over all disparities;
// wta_finish compares the count with the hits among the three exempt positions.
template <int N, bool SCHED>
__device__ __forceinline__ uint32_t wta_stage_count(const uint32_t (&s)[N], uint32_t key, UniqK uq) {
    const uint32_t T = wta_threshold(key >> 16, uq);
    const uint32_t Tc = min(T, 0x8000u), TT = Tc | (Tc << 16);
    // sign bits of two registers' differences gathered into one word (v_perm), counted by v_bcnt
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const uint32_t g = __builtin_amdgcn_perm(pk_sub(s[k], TT), pk_sub(s[k + 1], TT), 0x07050301u);
        cnt += (uint32_t)__builtin_popcount(g & 0x80808080u);
    }
    if (N & 1) cnt += (uint32_t)__builtin_popcount(pk_sub(s[N - 1], TT) & 0x80008000u);
    return SCHED ? row16_add_u32_s(cnt) : row16_add_u32(cnt);
}
// C: S[best-1] and S[best+1]: positions p and p+2 of the owning lane, 0 elsewhere, OR-reduced.
// Both have the parity of p and sit in registers kq = p >> 1 and kq + 1 (either may lie outside
// this lane: then the neighbouring lane supplies the value).
template <int N, bool SCHED>
__device__ __forceinline__ uint32_t wta_stage_neighbours(const uint32_t (&s)[N], uint32_t key, int q) {
    const int best = (int)(key & 0xFFFFu);
    const int p = best - 1 - q * 2 * N;  // position of best-1 in this lane's 2N values (may lie outside)
    const int kq = p >> 1;
    uint32_t vm = 0u, vp = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) {
        vm = kq == k ? s[k] : vm;
        vp = kq == k - 1 ? s[k] : vp;
    }
    // the low (p even) or high (p odd) halves of both: S[best-1] into the high half, S[best+1] into the low half
    const uint32_t nb = __builtin_amdgcn_perm(vm, vp, (p & 1) ? 0x07060302u : 0x05040100u);
    return SCHED ? row16_or_u32_s(nb) : row16_or_u32(nb);
}

template <int N>
__device__ __forceinline__ WtaPart wta_reduce(const uint32_t (&s)[N], int q, UniqK uq) {
    const uint32_t key = wta_stage_key<N, false>(s, q);
    const uint32_t cnt = wta_stage_count<N, false>(s, key, uq);
    const uint32_t nb = wta_stage_neighbours<N, false>(s, key, q);
    // best < 256 and the count only matters up to 4 (at most three positions are exempt): both fit the low half
    return WtaPart{key | (min(cnt, 255u) << 8), nb, 0u};
}

// The per-pixel arithmetic on the reduced words (lane-local: k_sweep runs it once per 16 pixels,
// each lane finishing a different pixel).  D = number of disparities.
// min_disp (EXT, cv::StereoSGBM minDisparity): the stored disparity is (best + min_disp) * 16 + sub-pixel, "none" is
// (min_disp - 1) * 16; the raw field stays the cost index (-1 = none).
// M2: the uniqueness test as "the smallest S outside best-1 .. best+1 is below T" (w.m2) instead of hits > exempt hits:
// the same predicate (some S outside the three exempt positions is below T).
template <int D, bool M2 = false>
__device__ __forceinline__ uint2 wta_finish(WtaPart w, UniqK uq, int min_disp = 0) {
    const int minS = (int)(w.key >> 16), best = (int)(w.key & 0xFFu);
    const uint32_t cnt = (w.key >> 8) & 0xFFu;
    const uint32_t T = wta_threshold((uint32_t)minS, uq);
    const int sm = (int)(w.nb >> 16), sp = (int)(w.nb & 0xFFFFu);
    const bool inner = 0 < best && best < D - 1;
    const uint32_t exempt = (uint32_t)((uint32_t)minS < T) + (uint32_t)(best > 0 && (uint32_t)sm < T) +
                            (uint32_t)(best < D - 1 && (uint32_t)sp < T);
    const bool bad = M2 ? w.m2 < T : cnt > exempt;
    // sub-pixel: ((sm - sp)*16 + den) / (2*den), C division (truncation toward zero), |quotient| <= 8.
    // In float without a correction step: floor(|num| / den2) = floor((2|num| + 1) / (2 den2)), whose
    // exact value is at least 1/(2 den2) >= 3.8e-6 away from an integer, while v_rcp_f32 + one
    // multiply are off by at most 8.5 * 1.5 * 2^-23 = 1.5e-6; all operands are exact in float (< 2^24).
    const int den = max(sm + sp - 2 * minS, 1), num = (sm - sp) * 16 + den;
    const int sgn = num >> 31;
    const uint32_t mag = (uint32_t)((num ^ sgn) - sgn);
    const uint32_t qf = (uint32_t)((float)(2u * mag + 1u) * __builtin_amdgcn_rcpf((float)(4 * den)));
    const int sub = inner ? (int)((qf ^ (uint32_t)sgn) - (uint32_t)sgn) : 0;
    const bool ok = minS < 32767 && !bad;  // minS == 32767: every S saturated, no minimum
    const int dv = ok ? (best + min_disp) * 16 + sub : (min_disp - 1) * 16, rw = ok ? best : -1;
    return make_uint2(((uint32_t)dv & 0xFFFFu) | ((uint32_t)rw << 16), (uint32_t)minS);
}

// Winner-take-all of one pixel by ONE lane, from the pixel's LDS slot (sv_sweep.hip.h, sweep_wta_words): slot[0 .. 16N) = S as
// 32 N halfwords in disparity order, slot[16N .. 16N + 16) = the 16 lane keys.  First minimum = the smallest key; S[best-1]
// and S[best+1] are two halfword reads; the uniqueness test needs the smallest S outside best-1 .. best+1: the (one or two)
// lanes that hold those positions are struck from the keys and the three positions from S (all ones), then it is the
// minimum of the remaining keys and of the two struck lanes' 2N values.  The slot is consumed (the row rewrites all of it).
template <int N>
__device__ __forceinline__ uint2 wta_from_slot(uint32_t* slot, UniqK uq, int min_disp) {
    constexpr int D = 32 * N;
    uint32_t k16[16];
    load_regs<16>(slot + 16 * N, k16);
    uint32_t key = k16[0];
#pragma unroll
    for (int i = 1; i < 16; i++) key = min(key, k16[i]);
    const int best = min((int)(key & 0xFFu), D - 1);   // (the clamp only matters for a slot nobody has written yet)
    const int bm = max(best - 1, 0), bp = min(best + 1, D - 1);
    u16_alias* const hw = reinterpret_cast<u16_alias*>(slot);
    const uint32_t nb = ((uint32_t)hw[bm] << 16) | (uint32_t)hw[bp];
    const int qa = bm / (2 * N), qb = bp / (2 * N);
    asm volatile("" ::: "memory");
    slot[16 * N + qa] = 0xFFFFFFFFu;
    slot[16 * N + qb] = 0xFFFFFFFFu;
    hw[bm] = (uint16_t)0xFFFFu;
    hw[best] = (uint16_t)0xFFFFu;
    hw[bp] = (uint16_t)0xFFFFu;
    asm volatile("" ::: "memory");    // the reads below stay behind the stores
    load_regs<16>(slot + 16 * N, k16);
    uint32_t ca[N], cb[N];
    load_regs<N>(slot + qa * N, ca);
    load_regs<N>(slot + qb * N, cb);
    uint32_t mo = k16[0];
#pragma unroll
    for (int i = 1; i < 16; i++) mo = min(mo, k16[i]);
    uint32_t mt = pk_min(ca[0], cb[0]);
#pragma unroll
    for (int k = 1; k < N; k++) mt = pk_min(mt, pk_min(ca[k], cb[k]));
    const uint32_t m2 = min(min(mt & 0xFFFFu, mt >> 16), mo >> 16);
    return wta_finish<D, true>(WtaPart{key, nb, m2}, uq, min_disp);
}

template <int N>
__device__ __forceinline__ uint2 wta_record(const uint32_t (&s)[N], int q, UniqK uq, int min_disp = 0) {
    return wta_finish<32 * N>(wta_reduce<N>(s, q, uq), uq, min_disp);
}

// ---------------------------------------------------------------------------------------------
// Stage 3: one SGM path.  Result of cost_aggregation_{lr,ud_lr,ud,rl_ud,rl}
// (sgbm_cuda/aggregation.cu:60-625) and, with dy = -1, of the second sweep of MODE_HH:
//   L(p,d) = C(p,d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, delta) - delta,  delta = min_d Lp + P2,
//   S(p,d) = min(S(p,d) + L(p,d), 32767)            (safe_add, aggregation.cu:4-10)
// A scanline ("chain") is 16 lanes; lane q holds disparities [q*2N, (q+1)*2N) as N packed u16
// pairs.  Horizontal paths: chain = image row.  Vertical / diagonal paths: chain = start column
// c, pixel at step i is (y_i, (c + dx*i) mod W1); when the column wraps the path restarts fresh
// (Lp = 0, delta = P2) exactly as the reference's diagonal kernels do at the image edge
// (aggregation.cu:233-256, :469-492) — and unlike them every pixel is visited for any aspect ratio.
// FIRST = this path initialises S (no read, no memset: bova_sgm.cu:106 is not needed).
// ---------------------------------------------------------------------------------------------
template <int N, int U>
struct AggBatch {
    uint32_t c[U][N], s[U][N];
    size_t off[U];
    uint32_t px[U];  // pixel index y*W + x + D of the step (LAST mode: where the WTA record goes)
    uint32_t rst;  // bit j: step j of the batch starts a fresh path; bit U: step 0 of the next batch does
};

// MODE bit 0 (FIRST): this path initialises S (no read).  MODE bit 1 (LAST): this path does not
// store S at all; the finished S of the pixel is still in registers, so WTA runs right here and
// only the 8-byte WTA record leaves the kernel (saves one volume write and one volume read).
template <int N, bool VERT, int MODE, int U>
__device__ __forceinline__ void agg_body(const uint32_t* __restrict__ C, uint32_t* __restrict__ S, int W1, int H, int dx, int dy,
                                         uint32_t P1P1, uint32_t P2P2, uint2* __restrict__ rec, int uniq, int W = 0, int col0 = 0,
                                         int min_disp = 0) {
    constexpr bool FIRST = (MODE & 1) != 0, LAST = (MODE & 2) != 0;  // W, col0, min_disp: only used by LAST (record index, WTA)
    const int q = threadIdx.x & 15;
    const int nchains = VERT ? W1 : H, nsteps = VERT ? H : W1;
    // Chains past the end repeat the last chain: same loads, same arithmetic, same stores from
    // lanes of the SAME wave (lock-step), so the duplicates are harmless and the loop stays
    // branch-free (a predicated store would split the body into blocks and make hipcc drain
    // vmcnt(0) at every block edge, which serialises the prefetch).
    const int chain = min((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), nchains - 1);
    const size_t pxN = (size_t)16 * N;  // u32 per pixel
    const uint32_t laneoff = (uint32_t)q * N;
    C += (size_t)blockIdx.y * H * W1 * pxN;  // frame of a batched launch
    S += (size_t)blockIdx.y * H * W1 * pxN;
    if (LAST) rec += (size_t)blockIdx.y * H * W;
    const UniqK uqk = make_uniq(uniq);

    // load cursor (runs one batch ahead of the compute)
    int lx, ly;
    if (VERT) { lx = chain; ly = dy > 0 ? 0 : H - 1; }
    else      { ly = chain; lx = dx > 0 ? 0 : W1 - 1; }

    auto issue = [&](AggBatch<N, U>& b) {
        uint32_t rs = 0;
#pragma unroll
        for (int j = 0; j < U; j++) {
            // clamped: a prefetch that runs past the last step reads a valid (unused) pixel
            const int cy = min(max(ly, 0), H - 1), cx = min(max(lx, 0), W1 - 1);
            b.off[j] = ((size_t)cy * W1 + cx) * pxN + laneoff;
            if (LAST) b.px[j] = (uint32_t)(cy * W + cx + col0);
            load_regs<N>(C + b.off[j], b.c[j]);
            if (!FIRST) load_regs<N>(S + b.off[j], b.s[j]);
            if (VERT) {
                ly += dy;
                lx += dx;
                const bool wrap = (lx >= W1) | (lx < 0);
                lx = lx >= W1 ? 0 : (lx < 0 ? W1 - 1 : lx);
                rs |= wrap ? (2u << j) : 0u;
            } else {
                lx += dx;
            }
        }
        b.rst = rs;
    };

    uint32_t r[N];
#pragma unroll
    for (int k = 0; k < N; k++) r[k] = 0;
    uint32_t dd = P2P2;

    auto step = [&](const uint32_t (&c)[N], const uint32_t (&s)[N], size_t off, uint32_t px, bool fresh) {
        if (VERT) {
            const uint32_t keep = fresh ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < N; k++) r[k] &= keep;
            dd = fresh ? P2P2 : dd;
        }
        const uint32_t prevHi = dpp<DPP_ROW_SHR1>(0x7FFF7FFFu, r[N - 1]);
        const uint32_t nextLo = dpp<DPP_ROW_SHL1>(0x7FFF7FFFu, r[0]);
        uint32_t m[N + 1];
        m[0] = __builtin_amdgcn_alignbit(r[0], prevHi, 16);
#pragma unroll
        for (int k = 1; k < N; k++) m[k] = __builtin_amdgcn_alignbit(r[k], r[k - 1], 16);
        m[N] = __builtin_amdgcn_alignbit(nextLo, r[N - 1], 16);
        uint32_t mn = 0xFFFFFFFFu;
        uint32_t snew[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            // C + min(a, dd) - dd  ==  C - max(dd - a, 0): one saturating subtract
            // plain 32-bit add / subtract where no half can carry or borrow (cheaper to issue than the packed
            // forms): every L <= 32767 and sv_check_params keeps 32767 + P1 <= 65535; C >= P2 >= max(dd - t, 0)
            const uint32_t t = pk_min(pk_min(m[k], m[k + 1]) + P1P1, r[k]);
            uint32_t L = c[k] - pk_sub_sat(dd, t);
            r[k] = L;
            mn = pk_min(mn, L);
            snew[k] = FIRST ? L : pk_add_sat15(s[k], L);
        }
        mn = pk_min(mn, __builtin_amdgcn_alignbit(mn, mn, 16));  // both halves = lane minimum
        mn = row16_min_u32(mn);
        dd = mn + P2P2;  // halves cannot carry: min + P2 < 65536
        if (LAST) rec[px] = wta_record<N>(snew, q, uqk, min_disp);  // all 16 lanes store the same record
        else store_regs<N>(S + off, snew);
    };

    uint32_t carry = 0;  // fresh-start flag for step 0 of the next batch
    auto compute = [&](AggBatch<N, U>& b) {
        const uint32_t flags = b.rst | carry;
#pragma unroll
        for (int j = 0; j < U; j++) step(b.c[j], b.s[j], b.off[j], b.px[j], (flags >> j) & 1u);
        carry = (b.rst >> U) & 1u;
    };

    // main loop: full double batches only, no branch inside
    const int nfull = nsteps - nsteps % (2 * U);
    AggBatch<N, U> A, B;
    if (nfull > 0) {
        issue(A);
        for (int i0 = 0; i0 < nfull; i0 += 2 * U) {
            issue(B);
            compute(A);
            issue(A);  // the last one is an unused, clamped prefetch
            compute(B);
        }
    }
    // tail (< 2U steps): closed-form position, no pipelining
    for (int i = nfull; i < nsteps; i++) {
        int x, y;
        bool fresh = false;
        if (VERT) {
            y = dy > 0 ? i : H - 1 - i;
            x = (chain + dx * i) % W1;
            x = x < 0 ? x + W1 : x;
            fresh = i > 0 && ((dx > 0 && x == 0) || (dx < 0 && x == W1 - 1));
        } else {
            y = chain;
            x = dx > 0 ? i : W1 - 1 - i;
        }
        const size_t off = ((size_t)y * W1 + x) * pxN + laneoff;
        uint32_t c[N], s[N];
        load_regs<N>(C + off, c);
        if (!FIRST) load_regs<N>(S + off, s);
        else {
#pragma unroll
            for (int k = 0; k < N; k++) s[k] = 0;
        }
        step(c, s, off, (uint32_t)(y * W + x + col0), fresh);
    }
}

template <int N, bool VERT, int MODE, int U>
__global__ __launch_bounds__(64) void k_agg(const uint32_t* __restrict__ C, uint32_t* __restrict__ S, int W1, int H, int dx,
                                             int dy, uint32_t P1P1, uint32_t P2P2, uint2* __restrict__ rec, int uniq, int W, int col0,
                                             int min_disp) {
    agg_body<N, VERT, MODE, U>(C, S, W1, H, dx, dy, P1P1, P2P2, rec, uniq, W, col0, min_disp);
}

// All paths of a small batch in ONE launch: blockIdx.z picks the path, every path writes its own L
// volume (FIRST mode).  A single 640x480 frame is then 8 x 128 waves in flight instead of 128 (the
// card has 1024 SIMDs; eight streams do not help, only four hardware queues exist per process).
struct PathSet {
    uint32_t* L[8];
    signed char dx[8], dy[8];
};
template <int N, int U>
__global__ __launch_bounds__(64) void k_agg_paths(const uint32_t* __restrict__ C, PathSet ps, int W1, int H, uint32_t P1P1,
                                                   uint32_t P2P2) {
    const int path = blockIdx.z;
    const int dx = ps.dx[path], dy = ps.dy[path];
    uint32_t* L = ps.L[path];
    if (dy != 0) {
        if ((int)(blockIdx.x * 4) >= W1) return;  // the grid is sized for the longer of the two chain counts
        agg_body<N, true, 1, U>(C, L, W1, H, dx, dy, P1P1, P2P2, nullptr, 0);
    } else {
        if ((int)(blockIdx.x * 4) >= H) return;
        agg_body<N, false, 1, U>(C, L, W1, H, dx, dy, P1P1, P2P2, nullptr, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 4: winner-take-all + uniqueness + sub-pixel.  Result of get_disparity
// (aggregation.cu:629-689).  16 lanes per pixel (same register layout as k_agg), full image width:
// columns [0, D) get the contract value -16 (the reference leaves them uninitialised).
// raw = -1 marks "no disparity" (reference: INVALID_DISP_SCALED in d_raw_disp, :642).
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_wta(const uint32_t* __restrict__ S, uint2* __restrict__ rec, int W, int H, int uniq, int col0,
                                              int W1, int min_disp) {
    const int q = threadIdx.x & 15;
    const size_t pid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const size_t npx = (size_t)W * H;
    const size_t p = min(pid, npx - 1);  // surplus groups repeat the last pixel (same value, same wave)
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const int xl = min(max(x - col0, 0), W1 - 1);
    S += (size_t)blockIdx.y * H * W1 * 16 * N;
    uint32_t s[N];
    load_regs<N>(S + ((size_t)y * W1 + xl) * 16 * N + q * N, s);
    uint2 r = wta_record<N>(s, q, make_uniq(uniq), min_disp);
    if (x < col0 || x >= col0 + W1) r = make_uint2(0xFFFF0000u | (uint32_t)(((min_disp - 1) * 16) & 0xFFFF), 32767u);  // {invalid, -1, 32767}
    rec[(size_t)blockIdx.y * npx + p] = r;
}

// Small batches: the paths run concurrently, each writing its own L volume (k_agg in FIRST mode on
// its own stream); this kernel adds them up with the saturation of safe_add (aggregation.cu:4-10:
// min(S + L, 32767) path by path = min(sum, 32767), all terms >= 0) and runs WTA on the sum.
struct PathBufs {
    const uint32_t* L[8];
    int n;
};
template <int N>
__global__ __launch_bounds__(256) void k_sum_wta(PathBufs pb, uint2* __restrict__ rec, int W, int H, int uniq, int col0, int W1,
                                                  int min_disp) {
    const int q = threadIdx.x & 15;
    const size_t pid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const size_t npx = (size_t)W1 * H;
    const size_t p = min(pid, npx - 1);  // surplus groups repeat the last pixel (same value, same wave)
    const int y = (int)(p / W1), xl = (int)(p - (size_t)y * W1);
    const size_t off = ((size_t)blockIdx.y * npx + p) * 16 * N + q * N;
    uint32_t acc[N], v[N];
    load_regs<N>(pb.L[0] + off, acc);
    for (int i = 1; i < pb.n; i++) {
        load_regs<N>(pb.L[i] + off, v);
#pragma unroll
        for (int k = 0; k < N; k++) acc[k] = pk_add_sat15(acc[k], v[k]);
    }
    rec[((size_t)blockIdx.y * H + y) * W + xl + col0] = wta_record<N>(acc, q, make_uniq(uniq), min_disp);
}

// ---------------------------------------------------------------------------------------------
// Stage 5: left-right check, one workgroup per image row.  Result of lrcheck
// (aggregation.cu:691-735): disp2/disp2cost are built right-to-left with a strict '>' (ties keep
// the larger x); the same winner is the minimum of the packed key (cost << 16 | 0xFFFF - x), so an
// LDS atomicMin reproduces it without the serial scan.
// ---------------------------------------------------------------------------------------------
// col0 / colN: the computable columns; min_disp (EXT): the raw field of a record is the cost index, the pixel it
// matches in the right image is x - index - min_disp, and "none" is (min_disp - 1) * 16.  The disparity the key
// stands for, x_src - x2, is the true one (index + min_disp), which is what cv::StereoSGBM keeps in disp2.
// One half of the consistency test: does disp2 at column xx disagree with dref by more than thresh?  k = the column's key
// (all ones: nobody matched into it).  cv::StereoSGBM leaves its scaled marker (minD - 1) * 16 in such an entry and tests
// "disp2 >= minD", which the marker passes for minD >= 2: the entry then counts as that disparity (oracle: svref_lrcheck_min).
__device__ __forceinline__ bool lr_disagrees(uint32_t k, int xx, int dref, int min_disp, int thresh) {
    int d2 = (0xFFFF - (int)(k & 0xFFFFu)) - xx;
    if (k == 0xFFFFFFFFu) {
        d2 = (min_disp - 1) * 16;
        if (d2 < min_disp) return false;
    }
    return abs(d2 - dref) > thresh;
}
__global__ __launch_bounds__(256) void k_lrcheck(const uint2* __restrict__ rec, int16_t* __restrict__ out, int W, int col0, int colN,
                                                  int min_disp, int thresh) {
    extern __shared__ uint32_t keys[];
    const int y = blockIdx.x + blockIdx.y * gridDim.x;  // gridDim.x = H, blockIdx.y = frame
    const uint2* rc = rec + (size_t)y * W;  // {disp16 | raw << 16, minS}; columns outside [col0, colN) are never read
    const int inv = (min_disp - 1) * 16;
    for (int x = threadIdx.x; x < W; x += blockDim.x) keys[x] = 0xFFFFFFFFu;
    __syncthreads();
    for (int x = col0 + threadIdx.x; x < colN; x += blockDim.x) {
        const uint2 r = rc[x];
        const int d = (int)(int16_t)(r.x >> 16);
        if (d >= 0) atomicMin(&keys[x - d - min_disp], (r.y << 16) | (uint32_t)(0xFFFF - x));
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int d1 = (x >= col0 && x < colN) ? (int)(int16_t)(rc[x].x & 0xFFFFu) : inv;
        if (d1 != inv) {
            int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            bool f1 = false, f2 = false;
            if (0 <= _x && _x < W) {
                uint32_t k = keys[_x];
                f1 = lr_disagrees(k, _x, _d, min_disp, thresh);
            }
            if (0 <= x_ && x_ < W) {
                uint32_t k = keys[x_];
                f2 = lr_disagrees(k, x_, d_, min_disp, thresh);
            }
            if (f1 && f2) d1 = inv;
        }
        out[(size_t)y * W + x] = (int16_t)d1;
    }
}

// ---------------------------------------------------------------------------------------------
// Stages 5 + 6 in one kernel (the production path when the median is on): a workgroup owns LRM_ROWS
// image rows, runs the left-right check of k_lrcheck for them and the row above and below, keeps the
// last three checked rows in LDS and takes the 3x3 median of k_median3 from there.  The checked map
// never goes to HBM and one launch replaces two (2.05 -> 1.25 ms per 1024 frames at 640x480).
// ---------------------------------------------------------------------------------------------
constexpr int LRM_ROWS = 16;
__device__ __forceinline__ void cswap16(int& a, int& b) { int t = min(a, b); b = max(a, b); a = t; }
__global__ __launch_bounds__(256) void k_lr_median(const uint2* __restrict__ rec, int16_t* __restrict__ out, int W, int H, int col0,
                                                    int colN, int min_disp, int thresh, int replicate) {
    extern __shared__ uint32_t lrm_lds[];
    uint32_t* keys = lrm_lds;                                       // [W]
    int16_t* rows = reinterpret_cast<int16_t*>(lrm_lds + W);        // [3][W]: checked rows yl-2, yl-1, yl (slot = (yl + 3) % 3)
    const int ya = blockIdx.x * LRM_ROWS, yb = min(ya + LRM_ROWS, H);
    const int inv = (min_disp - 1) * 16;
    rec += (size_t)blockIdx.y * W * H;
    out += (size_t)blockIdx.y * W * H;
    for (int yl = ya - 1; yl <= yb; yl++) {
        const int yc = min(max(yl, 0), H - 1);   // above / below the image: the edge row again (replicate border)
        const uint2* rc = rec + (size_t)yc * W;
        int16_t* lr = rows + (size_t)((yl + 3) % 3) * W;
        for (int x = threadIdx.x; x < W; x += blockDim.x) keys[x] = 0xFFFFFFFFu;
        __syncthreads();
        for (int x = col0 + threadIdx.x; x < colN; x += blockDim.x) {
            const uint2 r = rc[x];
            const int d = (int)(int16_t)(r.x >> 16);
            if (d >= 0) atomicMin(&keys[x - d - min_disp], (r.y << 16) | (uint32_t)(0xFFFF - x));
        }
        __syncthreads();
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            int d1 = (x >= col0 && x < colN) ? (int)(int16_t)(rc[x].x & 0xFFFFu) : inv;
            if (d1 != inv) {
                int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
                int _x = x - _d, x_ = x - d_;
                bool f1 = false, f2 = false;
                if (0 <= _x && _x < W) {
                    uint32_t k = keys[_x];
                    f1 = lr_disagrees(k, _x, _d, min_disp, thresh);
                }
                if (0 <= x_ && x_ < W) {
                    uint32_t k = keys[x_];
                    f2 = lr_disagrees(k, x_, d_, min_disp, thresh);
                }
                if (f1 && f2) d1 = inv;
            }
            lr[x] = (int16_t)d1;
        }
        __syncthreads();
        const int y = yl - 1;   // its three rows are in LDS now
        if (y < ya) continue;
        const int16_t* r0 = rows + (size_t)((y - 1 + 3) % 3) * W;
        const int16_t* r1 = rows + (size_t)((y + 3) % 3) * W;
        const int16_t* r2 = rows + (size_t)((y + 1 + 3) % 3) * W;
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            const bool border = (x < 1 || y < 1 || x >= W - 1 || y >= H - 1);
            int res;
            if (border && !replicate) {
                res = r1[x];
            } else {
                const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
                int v[9] = {r0[xl], r0[x], r0[xr], r1[xl], r1[x], r1[xr], r2[xl], r2[x], r2[xr]};
                cswap16(v[1], v[2]); cswap16(v[4], v[5]); cswap16(v[7], v[8]);
                cswap16(v[0], v[1]); cswap16(v[3], v[4]); cswap16(v[6], v[7]);
                cswap16(v[1], v[2]); cswap16(v[4], v[5]); cswap16(v[7], v[8]);
                cswap16(v[0], v[3]); cswap16(v[5], v[8]); cswap16(v[4], v[7]);
                cswap16(v[3], v[6]); cswap16(v[1], v[4]); cswap16(v[2], v[5]);
                cswap16(v[4], v[7]); cswap16(v[4], v[2]); cswap16(v[6], v[4]);
                cswap16(v[4], v[2]);
                res = v[4];
            }
            out[(size_t)y * W + x] = (int16_t)res;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 6: 3x3 median.  Result of MedianFilter (aggregation.cu:737-768); replicate = 1 gives
// cv::medianBlur's border handling instead of copying the border.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(int& a, int& b) { int t = min(a, b); b = max(a, b); a = t; }
__global__ __launch_bounds__(256) void k_median3(const int16_t* __restrict__ in, int16_t* __restrict__ out, int W, int H,
                                                  int replicate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    in += (size_t)blockIdx.z * W * H;
    out += (size_t)blockIdx.z * W * H;
    const bool border = (x < 1 || y < 1 || x >= W - 1 || y >= H - 1);
    if (border && !replicate) { out[(size_t)y * W + x] = in[(size_t)y * W + x]; return; }
    int v[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int yy = min(max(y + i - 1, 0), H - 1), xx = min(max(x + j - 1, 0), W - 1);
            v[i * 3 + j] = in[(size_t)yy * W + xx];
        }
    // median-of-9 exchange network
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    out[(size_t)y * W + x] = (int16_t)v[4];
}

// ---------------------------------------------------------------------------------------------
// Speckle filter.  Result of cv::filterSpeckles (inside cv::StereoSGBM::compute, which matchSGBM
// calls, StereoMatcher.cpp:412): 4-connected components of pixels != newVal joined when
// |delta| <= maxDiff; components of size <= maxSize become newVal.  Union-find in global memory
// (label = smallest pixel index of the component), so the result is order-independent.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* __restrict__ lab, int a) {
    int p = lab[a];
    while (p != a) { a = p; p = lab[a]; }
    return a;
}
__device__ __forceinline__ void uf_union(int* __restrict__ lab, int a, int b) {
    while (true) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        int old = atomicMin(&lab[b], a);
        if (old == b) return;
        b = old;
    }
}
// Frames of a batch are a grid dimension (blockIdx.y, or .z for the 2-D merge kernel); n = pixels per
// frame, labels are pixel indices within the frame.
__global__ __launch_bounds__(256) void k_speckle_init(int* __restrict__ lab, int* __restrict__ cnt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t fo = (size_t)blockIdx.y * n;
    lab[fo + i] = i;
    cnt[fo + i] = 0;
}
__global__ __launch_bounds__(256) void k_speckle_merge(const int16_t* __restrict__ img, int* __restrict__ lab, int W, int H,
                                                        int new_val, int max_diff) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t fo = (size_t)blockIdx.z * H * W;
    img += fo;
    lab += fo;
    int i = y * W + x, v = img[i];
    if (v == new_val) return;
    if (x + 1 < W) { int u = img[i + 1]; if (u != new_val && abs(u - v) <= max_diff) uf_union(lab, i, i + 1); }
    if (y + 1 < H) { int u = img[i + W]; if (u != new_val && abs(u - v) <= max_diff) uf_union(lab, i, i + W); }
}
// Component sizes.  A large component would otherwise take one global atomic per pixel on a single
// address (measured 2.4 ms per 640x480 frame); pixels of a wave mostly share a root, so each wave
// first groups its lanes by root (readfirstlane / ballot loop) and the block then merges the wave
// totals of equal roots in LDS, leaving a handful of global atomics per block.
__global__ __launch_bounds__(1024) void k_speckle_count(const int16_t* __restrict__ img, int* __restrict__ lab,
                                                         int* __restrict__ cnt, int n, int new_val) {
    constexpr int MAXE = 128;
    __shared__ int e_root[MAXE], e_cnt[MAXE], e_n;
    if (threadIdx.x == 0) e_n = 0;
    __syncthreads();
    const size_t fo = (size_t)blockIdx.y * n;
    img += fo;
    lab += fo;
    cnt += fo;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n && img[i] != new_val;
    int r = -1;
    if (active) {
        r = uf_find(lab, i);
        lab[i] = r;  // path compression: only ever lowers a label towards its root
    }
    while (__ballot(active)) {
        const unsigned long long live = __ballot(active);
        const int leader = __ffsll((long long)live) - 1;
        const int r0 = __shfl(r, leader);
        const bool mine = active && r == r0;
        const unsigned long long same = __ballot(mine);
        if ((int)(threadIdx.x & 63) == leader) {
            const int slot = atomicAdd(&e_n, 1);
            if (slot < MAXE) { e_root[slot] = r0; e_cnt[slot] = __popcll(same); }
            else atomicAdd(&cnt[r0], __popcll(same));
        }
        active = active && !mine;
    }
    __syncthreads();
    const int ne = min(e_n, MAXE), t = threadIdx.x;
    int first = t;
    if (t < ne)
        for (int j = 0; j < t; j++)
            if (e_root[j] == e_root[t]) { first = j; break; }
    __syncthreads();
    if (t < ne && first != t) atomicAdd(&e_cnt[first], e_cnt[t]);
    __syncthreads();
    if (t < ne && first == t) atomicAdd(&cnt[e_root[t]], e_cnt[t]);
}
__global__ __launch_bounds__(256) void k_speckle_apply(int16_t* __restrict__ img, const int* __restrict__ lab,
                                                        const int* __restrict__ cnt, int n, int new_val, int max_size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t fo = (size_t)blockIdx.y * n;
    if (i >= n || img[fo + i] == new_val) return;
    lab += fo;
    int r = lab[i];
    while (lab[r] != r) r = lab[r];
    if (cnt[fo + r] <= max_size) img[fo + i] = (int16_t)new_val;
}

// ---------------------------------------------------------------------------------------------
// Per-pixel helpers around the matcher.
// ---------------------------------------------------------------------------------------------
// cv::cvtColor(CV_BGR2GRAY) on RGB-ordered bytes (StereoMatcher.cpp:429-430).
__global__ __launch_bounds__(256) void k_gray(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ gray, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a = rgb[3 * i], b = rgb[3 * i + 1], c = rgb[3 * i + 2];
    gray[i] = (uint8_t)((a * 1868u + b * 9617u + c * 4899u + (1u << 13)) >> 14);
}
// cv::ximgproc::getDisparityVis (StereoMatcher.cpp:437).
__global__ __launch_bounds__(256) void k_vis(const int16_t* __restrict__ d16, uint8_t* __restrict__ vis, size_t n, int scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v = (int)d16[i] * scale;
    int qv = v >> 4, r = v & 15;
    if (r > 8 || (r == 8 && (qv & 1))) qv++;
    vis[i] = (uint8_t)min(max(qv, 0), 255);
}
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on 8UC3 (stereoCamera.cpp:573-574): a gather with
// coalesced destination rows; the source neighbourhood of a row is a narrow band, served by L2.
__global__ __launch_bounds__(256) void k_remap_rgb(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                    const float* __restrict__ mapx, const float* __restrict__ mapy, int W,
                                                    int H) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    src += (size_t)blockIdx.z * W * H * 3;  // frames of a batch share the camera's maps
    dst += (size_t)blockIdx.z * W * H * 3;
    const size_t o = (size_t)y * W + x;
    float fxm = fminf(fmaxf(mapx[o] * 32.0f, -1.0e9f), 1.0e9f);
    float fym = fminf(fmaxf(mapy[o] * 32.0f, -1.0e9f), 1.0e9f);
    int sx = __float2int_rn(fxm), sy = __float2int_rn(fym);
    int ix = min(max(sx >> 5, -32768), 32767), iy = min(max(sy >> 5, -32768), 32767);
    int fx = sx & 31, fy = sy & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const uint8_t* p00 = src + ((size_t)iy * W + ix) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int a = (y0 && x0) ? p00[c] : 0, b = (y0 && x1) ? p00[3 + c] : 0;
        int e = (y1 && x0) ? p00[(size_t)W * 3 + c] : 0, f = (y1 && x1) ? p00[(size_t)W * 3 + 3 + c] : 0;
        int v = (a * w00 + b * w01 + e * w10 + f * w11 + (1 << 14)) >> 15;
        dst[o * 3 + c] = (uint8_t)min(v, 255);
    }
}
// DispModule::depthFromDisparity (DispModule.cpp:320-366), non-aliased read.
struct DepthCoef { float r02, r12, r22, q00, q03, q11, q13, q23, q32, q33; };
__global__ __launch_bounds__(256) void k_depth(const int16_t* __restrict__ d16, const float2* __restrict__ mapper,
                                                float* __restrict__ depth, int W, int H, DepthCoef c) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (v >= W) return;
    d16 += (size_t)blockIdx.z * W * H;      // frames of a batch share the mapper
    depth += (size_t)blockIdx.z * W * H;
    const size_t o = (size_t)u * W + v;
    float2 m = mapper[o];
    float usign = rintf(m.x), vsign = rintf(m.y);
    float out = 0.0f;
    if (!(usign < 0 || usign >= (float)W || vsign < 0 || vsign >= (float)H)) {
        float dsp = (float)d16[(size_t)vsign * W + (size_t)usign] * 0.0625f;
        float num = c.r02 * (vsign * c.q00 + c.q03) + c.r12 * (usign * c.q11 + c.q13) + c.r22 * c.q23;
        out = num / (dsp * c.q32 + c.q33);
    }
    depth[o] = out;
}

// SFM::fillWorld3D (modules/SFM/SFM.cpp:1284-1347) with the point arithmetic of
// DispModule::get3DPoints "ROOT" (DispModule.cpp:1103-1160): cartesian and cylindrical world images
// of the ORIGINAL left view, FP64 like the reference.  Hr = first three rows of
// HL_root * [RLrect^T 0; 0 1].  Pixels that map outside the rectified image, or whose z is not in
// [0, 10] m, are 0 (the reference zeroes both images first).
struct WorldCoef { double q00, q03, q11, q13, q23, q32, q33, Hr[12]; };
__global__ __launch_bounds__(256) void k_world_image(const int16_t* __restrict__ d16, const float2* __restrict__ mapper,
                                                      float* __restrict__ cart, float* __restrict__ cyl, int W, int H, WorldCoef c) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= W) return;
    const size_t o = (size_t)v * W + u;
    const float2 m = mapper[o];
    const float ur = rintf(m.x), vr = rintf(m.y);
    float cx = 0.f, cy = 0.f, cz = 0.f, rr = 0.f, th = 0.f;
    if (!(ur < 0 || ur >= (float)W || vr < 0 || vr >= (float)H)) {  // also false for NaN coordinates
        double x = ((double)m.x + 1) * c.q00 + c.q03;
        double y = ((double)m.y + 1) * c.q11 + c.q13;
        double z = c.q23;
        const double disparity = (double)d16[(size_t)vr * W + (size_t)ur] / 16.0;
        const double w = disparity * c.q32 + c.q33;
        x /= w; y /= w; z /= w;
        if (!(z > 10.0 || z < 0.0)) {
            const double X = c.Hr[0] * x + c.Hr[1] * y + c.Hr[2] * z + c.Hr[3];
            const double Y = c.Hr[4] * x + c.Hr[5] * y + c.Hr[6] * z + c.Hr[7];
            const double Z = c.Hr[8] * x + c.Hr[9] * y + c.Hr[10] * z + c.Hr[11];
            cx = (float)X; cy = (float)Y; cz = (float)Z;
            rr = (float)sqrt(X * X + Y * Y);
            th = (float)atan2(Y, X);
        }
    }
    cart[3 * o] = cx; cart[3 * o + 1] = cy; cart[3 * o + 2] = cz;
    cyl[3 * o] = rr; cyl[3 * o + 1] = th; cyl[3 * o + 2] = cz;
}

// ---------------------------------------------------------------------------------------------
// WLS post-filter.  Result of wls_filter->filter(...) as StereoMatcherNew::filterWLS calls it
// (StereoMatcher.cpp:154-160) on a filter made by createDisparityWLSFilterGeneric(false) (:265):
// outside the ROI of computeROI2 (:12-29) the output is 16*(min_disp-1) = -16, inside it is the
// fast global smoother (3 iterations, lambda x0.25 per iteration) of the disparity guided by the
// rectified left RGB image.  Each iteration is one tridiagonal (Thomas) solve per row and one per
// column.  The solves keep the sequential float operation order of the CPU statement, so results
// match it bit for bit; parallelism comes from rows x images (disp16 and its 8-bit
// visualisation are filtered together, frames of a batch too).
//   cur / tmp : [n_img][rh][rw] float   (n_img = n_frames * K, image i uses the weights of frame i / K)
//   wh / wv   : [n_frames][rh][rw] float, negative edge weights, 0 on the last column / row
// ---------------------------------------------------------------------------------------------
constexpr int WLS_LUT_SIZE = 3 * 256 * 256;

__global__ __launch_bounds__(256) void k_wls_weights(const uint8_t* __restrict__ guide, const float* __restrict__ lut,
                                                      float* __restrict__ wh, float* __restrict__ wv, int W, int H, int rx,
                                                      int ry, int rw, int rh) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, f = blockIdx.z;
    if (j >= rw) return;
    const uint8_t* p = guide + ((size_t)f * H * W + (size_t)(ry + i) * W + rx + j) * 3;
    const int a0 = p[0], a1 = p[1], a2 = p[2];
    float h = 0.0f, v = 0.0f;
    if (j < rw - 1) {
        const int d0 = a0 - p[3], d1 = a1 - p[4], d2 = a2 - p[5];
        h = lut[d0 * d0 + d1 * d1 + d2 * d2];
    }
    if (i < rh - 1) {
        const uint8_t* q = p + (size_t)W * 3;
        const int d0 = a0 - q[0], d1 = a1 - q[1], d2 = a2 - q[2];
        v = lut[d0 * d0 + d1 * d1 + d2 * d2];
    }
    const size_t o = ((size_t)f * rh + i) * rw + j;
    wh[o] = h;
    wv[o] = v;
}

// disparity16 (k = 0) and the 8-bit visualisation (k = 1, when K == 2) of the ROI -> float
__global__ __launch_bounds__(256) void k_wls_load(const int16_t* __restrict__ d16, const uint8_t* __restrict__ vis,
                                                   float* __restrict__ cur, int W, int H, int rx, int ry, int rw, int rh, int K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, img = blockIdx.z;
    if (j >= rw) return;
    const int f = img / K, k = img - f * K;
    const size_t src = (size_t)f * H * W + (size_t)(ry + i) * W + rx + j;
    cur[((size_t)img * rh + i) * rw + j] = k == 0 ? (float)d16[src] : (float)vis[src];
}

// float -> int16 / u8 with saturate_cast(cvRound(.)), -16 outside the ROI (0 after the u8 saturation)
__global__ __launch_bounds__(256) void k_wls_store(const float* __restrict__ cur, int16_t* __restrict__ out16,
                                                    uint8_t* __restrict__ out8, int W, int H, int rx, int ry, int rw, int rh,
                                                    int K) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
    if (x >= W) return;
    const int f = img / K, k = img - f * K;
    const bool in = x >= rx && x < rx + rw && y >= ry && y < ry + rh;
    const float v = in ? cur[((size_t)img * rh + (y - ry)) * rw + (x - rx)] : -16.0f;
    const float r = rintf(v);  // round half to even
    const size_t o = (size_t)f * H * W + (size_t)y * W + x;
    if (k == 0) out16[o] = (int16_t)fminf(fmaxf(r, -32768.0f), 32767.0f);
    else out8[o] = (uint8_t)fminf(fmaxf(r, 0.0f), 255.0f);
}

// Row solves: one lane per row, the row walked four samples (one 16-byte access) at a time, FGS_PD such groups
// of loads in flight ahead of the dependent solve chain (mul, sub, div per sample: ~45 cycles each), so the
// memory latency hides behind the arithmetic.  A wave's 64 lanes touch 64 different rows per access; the eight
// accesses that share a 128-byte line follow each other closely and are served by L1 / L2 (HBM sees each line
// once).  (The first form staged 64x64 tiles through LDS with coalesced accesses; its load / solve / store
// phases ran one after the other and exposed the memory latency 16 times per tile: 2.3 x the column pass.)
// The forward sweep leaves the modified right-hand side in cur and the modified super-diagonal in tmp; the
// backward sweep walks the row in reverse.  Row starts are only 4-byte aligned (ROI widths are arbitrary): the
// 16-byte global accesses of gfx950 need no more.
constexpr int FGS_T = 64, FGS_PD = 4;
__device__ __forceinline__ void fgs_step(float w, float& c, float& t, float lambda, float& coef_prev, float& prev_cur,
                                         float& prev_tmp) {
    const float coef_cur = lambda * w;
    const float denom = (1.0f - coef_prev - coef_cur) - coef_prev * prev_tmp;
    prev_tmp = coef_cur / denom;
    prev_cur = (c - coef_prev * prev_cur) / denom;
    t = prev_tmp;
    c = prev_cur;
    coef_prev = coef_cur;
}
__device__ __forceinline__ void fgs_ld4(const float* __restrict__ p, float (&v)[4]) {
    uint32_t r[4];
    load_regs<4>(reinterpret_cast<const uint32_t*>(p), r);
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = __uint_as_float(r[u]);
}
__device__ __forceinline__ void fgs_st4(float* __restrict__ p, const float (&v)[4]) {
    uint32_t r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) r[u] = __float_as_uint(v[u]);
    store_regs<4>(reinterpret_cast<uint32_t*>(p), r);
}
__global__ __launch_bounds__(64) void k_fgs_hpass(float* __restrict__ cur, const float* __restrict__ wh, float* __restrict__ tmp,
                                                   int rw, int rh, int K, float lambda) {
    const int img = blockIdx.y, f = img / K;
    const int row = blockIdx.x * FGS_T + threadIdx.x;
    // No barrier in this kernel: lanes past the last row leave, and every store below is unconditional.
    if (row >= rh) return;
    float* curR = cur + ((size_t)img * rh + row) * rw;
    float* tmpR = tmp + ((size_t)img * rh + row) * rw;
    const float* whR = wh + ((size_t)f * rh + row) * rw;
    const int ng = rw / 4;        // full groups of four samples; the last rw % 4 samples go one by one
    const int nblk = ng / FGS_PD; // blocks of FGS_PD groups, each group with registers of its own; the last ng % FGS_PD groups go one by one
    // How the prefetch survives the compiler: a wait for a load is emitted as "all but the n youngest vector memory
    // operations done", with n counted along every path into the wait and the smallest taken.  So (1) no load or store
    // of the loops sits behind a condition, (2) the groups in flight keep their registers (a rotation by copies would
    // have to wait for the loads it copies), and (3) the loop is entered with at least as many operations behind the
    // first group's loads as its back edge carries: the prologue stores every group it loaded back unchanged.
    float prev_cur = 0.0f, prev_tmp = 0.0f, coef_prev = 0.0f;
    if (nblk > 0) {
        float c[FGS_PD][4], w[FGS_PD][4];
#pragma unroll
        for (int p = 0; p < FGS_PD; p++) { fgs_ld4(curR + 4 * p, c[p]); fgs_ld4(whR + 4 * p, w[p]); }
#pragma unroll
        for (int p = 0; p < FGS_PD; p++) { fgs_st4(curR + 4 * p, c[p]); fgs_st4(tmpR + 4 * p, w[p]); }   // cur unchanged, tmp is scratch
        for (int blk = 0; blk < nblk; blk++) {
#pragma unroll
            for (int p = 0; p < FGS_PD; p++) {
                const int g = blk * FGS_PD + p;
                float tt[4];
#pragma unroll
                for (int u = 0; u < 4; u++) fgs_step(w[p][u], c[p][u], tt[u], lambda, coef_prev, prev_cur, prev_tmp);
                fgs_st4(curR + 4 * g, c[p]);
                fgs_st4(tmpR + 4 * g, tt);
                const int gn = min(g + FGS_PD, ng - 1);   // past the last block: an unused, in-bounds prefetch
                fgs_ld4(curR + 4 * gn, c[p]);
                fgs_ld4(whR + 4 * gn, w[p]);
            }
        }
    }
    for (int g = nblk * FGS_PD; g < ng; g++) {
        float cc[4], ww[4], tt[4];
        fgs_ld4(curR + 4 * g, cc);
        fgs_ld4(whR + 4 * g, ww);
#pragma unroll
        for (int u = 0; u < 4; u++) fgs_step(ww[u], cc[u], tt[u], lambda, coef_prev, prev_cur, prev_tmp);
        fgs_st4(curR + 4 * g, cc);
        fgs_st4(tmpR + 4 * g, tt);
    }
    for (int j = 4 * ng; j < rw; j++) {
        float cc = curR[j], tt;
        fgs_step(whR[j], cc, tt, lambda, coef_prev, prev_cur, prev_tmp);
        curR[j] = cc;
        tmpR[j] = tt;
    }
    // this lane reads back what it has just written
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // backward: the last sample of the row is final after the forward sweep
    float next = prev_cur;
    for (int j = rw - 2; j >= 4 * ng; j--) {
        next = curR[j] - tmpR[j] * next;
        curR[j] = next;
    }
    auto back4 = [&](int g, float (&cc)[4], const float (&tt)[4]) {
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            next = (4 * g + u == rw - 1) ? cc[u] : cc[u] - tt[u] * next;
            cc[u] = next;
        }
    };
    for (int g = ng - 1; g >= nblk * FGS_PD; g--) {
        float cc[4], tt[4];
        fgs_ld4(curR + 4 * g, cc);
        fgs_ld4(tmpR + 4 * g, tt);
        back4(g, cc, tt);
        fgs_st4(curR + 4 * g, cc);
    }
    if (nblk > 0) {
        float c[FGS_PD][4], t[FGS_PD][4];
        const int gtop = nblk * FGS_PD - 1;
#pragma unroll
        for (int p = 0; p < FGS_PD; p++) { fgs_ld4(curR + 4 * (gtop - p), c[p]); fgs_ld4(tmpR + 4 * (gtop - p), t[p]); }
#pragma unroll
        for (int p = 0; p < FGS_PD; p++) fgs_st4(curR + 4 * (gtop - p), c[p]);   // unchanged
        for (int blk = 0; blk < nblk; blk++) {
#pragma unroll
            for (int p = 0; p < FGS_PD; p++) {
                const int g = gtop - blk * FGS_PD - p;
                back4(g, c[p], t[p]);
                fgs_st4(curR + 4 * g, c[p]);
                const int gn = max(g - FGS_PD, 0);
                fgs_ld4(curR + 4 * gn, c[p]);
                fgs_ld4(tmpR + 4 * gn, t[p]);
            }
        }
    }
}

// The row pass for LARGE launches (many frames: hundreds of waves per CU over the launch): the lane-per-row form
// above issues 64 different cache lines per access, which the texture-address path of a busy CU cannot keep
// up with (1.8 ms per 128-frame pass against 0.86 ms); this form moves 64x64 tiles between HBM / L2 and LDS with
// coalesced accesses and lets the solving lane read along its own LDS row (stride 65 words: conflict-free), the
// other waves of the CU covering its load / solve / store phases.  Inside a tile the lane moves FGS_R samples at a
// time between LDS and registers.  Two tiles of LDS per wave (the super-diagonal takes the weights' place).
constexpr int FGS_R = 16;
// 64 x TC tile <-> global: a wave-load covers 64 / TC rows of TC contiguous floats (256 or 128 bytes each), 8 loads in flight.
// Rows past nrows re-read the last valid row (never written back); columns past nc are skipped.
// TC = 32 (round 3): two tiles take 17 KB instead of 33 KB, so a CU holds 9 of these one-wave workgroups instead of 4 - the
// kernel lives on other waves covering a wave's load / solve / store phases.  WLS of a 16-bit map + its 8-bit view, 640x480, per frame
// of a 128- or 1024-frame call: 29.1 us with 64-column tiles, 25.1 with 32, 32.4 with 16 (64-byte row segments).
template <int TC>
__device__ __forceinline__ void fgs_tile_load(float (*t)[TC + 1], const float* __restrict__ g, int rw, int nrows, int c0, int nc, int lane) {
    constexpr int SUB = FGS_T / TC;
    const int col = lane % TC, sub = lane / TC;
    if (col >= nc) return;
#pragma unroll 1
    for (int r = 0; r < FGS_T; r += 8 * SUB) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = g[(size_t)min(r + u * SUB + sub, nrows - 1) * rw + c0 + col];
#pragma unroll
        for (int u = 0; u < 8; u++) t[r + u * SUB + sub][col] = a[u];
    }
}
template <int TC>
__device__ __forceinline__ void fgs_tile_store(const float (*t)[TC + 1], float* __restrict__ g, int rw, int nrows, int c0, int nc, int lane) {
    constexpr int SUB = FGS_T / TC;
    const int col = lane % TC, sub = lane / TC;
    if (col >= nc) return;
    for (int r = sub; r < nrows; r += SUB) g[(size_t)r * rw + c0 + col] = t[r][col];
}
template <int TC>
__global__ __launch_bounds__(64) void k_fgs_hpass_tiles(float* __restrict__ cur, const float* __restrict__ wh, float* __restrict__ tmp,
                                                   int rw, int rh, int K, float lambda) {
    static_assert(FGS_T % TC == 0 && TC % FGS_R == 0, "tile shape");
    __shared__ float t_c[FGS_T][TC + 1], t_w[FGS_T][TC + 1];   // t_w: the weights on the way in, the super-diagonal on the way out / back in
    float (*const t_t)[TC + 1] = t_w;
    const int img = blockIdx.y, f = img / K, lane = threadIdx.x;
    const int r0 = blockIdx.x * FGS_T, nrows = min(FGS_T, rh - r0);
    float* curI = cur + ((size_t)img * rh + r0) * rw;
    float* tmpI = tmp + ((size_t)img * rh + r0) * rw;
    const float* whI = wh + ((size_t)f * rh + r0) * rw;
    float prev_cur = 0.0f, prev_tmp = 0.0f, coef_prev = 0.0f;
    const int nchunks = (rw + TC - 1) / TC;
    for (int c = 0; c < nchunks; c++) {
        const int c0 = c * TC, nc = min(TC, rw - c0);
        fgs_tile_load<TC>(t_c, curI, rw, nrows, c0, nc, lane);
        fgs_tile_load<TC>(t_w, whI, rw, nrows, c0, nc, lane);
        __syncthreads();
        // rows past nrows solve garbage that is never written back
        for (int j0 = 0; j0 < nc; j0 += FGS_R) {
            float cc[FGS_R], ww[FGS_R], tt[FGS_R];
#pragma unroll
            for (int u = 0; u < FGS_R; u++) { cc[u] = t_c[lane][j0 + u]; ww[u] = t_w[lane][j0 + u]; }
#pragma unroll
            for (int u = 0; u < FGS_R; u++) {
                tt[u] = 0.0f;
                if (j0 + u < nc) fgs_step(ww[u], cc[u], tt[u], lambda, coef_prev, prev_cur, prev_tmp);
            }
#pragma unroll
            for (int u = 0; u < FGS_R; u++) { t_c[lane][j0 + u] = cc[u]; t_t[lane][j0 + u] = tt[u]; }
        }
        __syncthreads();
        fgs_tile_store<TC>(t_c, curI, rw, nrows, c0, nc, lane);
        fgs_tile_store<TC>(t_t, tmpI, rw, nrows, c0, nc, lane);
        __syncthreads();
    }
    float next = 0.0f;
    for (int c = nchunks - 1; c >= 0; c--) {
        const int c0 = c * TC, nc = min(TC, rw - c0);
        fgs_tile_load<TC>(t_c, curI, rw, nrows, c0, nc, lane);
        fgs_tile_load<TC>(t_t, tmpI, rw, nrows, c0, nc, lane);
        __syncthreads();
        for (int j0 = TC - FGS_R; j0 >= 0; j0 -= FGS_R) {
            float cc[FGS_R], tt[FGS_R];
#pragma unroll
            for (int u = 0; u < FGS_R; u++) { cc[u] = t_c[lane][j0 + u]; tt[u] = t_t[lane][j0 + u]; }
#pragma unroll
            for (int u = FGS_R - 1; u >= 0; u--) {
                const int j = j0 + u;
                if (j < nc) {
                    // the last sample of the row is final after the forward sweep
                    next = (c0 + j == rw - 1) ? cc[u] : cc[u] - tt[u] * next;
                    cc[u] = next;
                }
            }
#pragma unroll
            for (int u = 0; u < FGS_R; u++) t_c[lane][j0 + u] = cc[u];
        }
        __syncthreads();
        fgs_tile_store<TC>(t_c, curI, rw, nrows, c0, nc, lane);
        __syncthreads();
    }
}

// Column solves: one lane per column, rows walked in order (coalesced across the wave), FGS_V rows per trip with
// the loads of FGS_VD trips in flight ahead of the solve chain.  As in k_fgs_hpass the trips in flight keep their
// registers, no load or store of the main loops sits behind a condition, and the prologue stores what it loaded back
// unchanged, so that the compiler's wait counts keep the prefetch alive (see there).
constexpr int FGS_V = 8, FGS_VD = 3;   // (8, 2), (4, 3), (4, 6), (6, 4) and FGS_PD 2 .. 8 measured within 6 % of this
__global__ __launch_bounds__(64) void k_fgs_vpass(float* __restrict__ cur, const float* __restrict__ wv, float* __restrict__ tmp,
                                                   int rw, int rh, int K, float lambda) {
    const int j = min((int)(blockIdx.x * blockDim.x + threadIdx.x), rw - 1);  // surplus lanes repeat the last column (same wave: same values)
    const int img = blockIdx.y, f = img / K;
    float* curI = cur + (size_t)img * rh * rw + j;
    float* tmpI = tmp + (size_t)img * rh * rw + j;
    const float* wvI = wv + (size_t)f * rh * rw + j;
    const int ntrip = rh / FGS_V;        // full trips of FGS_V rows; the last rh % FGS_V rows go one by one
    const int nblk = ntrip / FGS_VD;     // blocks of FGS_VD trips with registers of their own; the other full trips go one by one
    float prev_cur = 0.0f, prev_tmp = 0.0f, coef_prev = 0.0f;
    auto ld_trip = [&](const float* a, const float* b2, int t, float (&x)[FGS_V], float (&y)[FGS_V]) {
#pragma unroll
        for (int u = 0; u < FGS_V; u++) {
            const size_t o = (size_t)(t * FGS_V + u) * rw;
            x[u] = a[o];
            y[u] = b2[o];
        }
    };
    auto fwd_trip = [&](int t, float (&c)[FGS_V], const float (&w)[FGS_V]) {
#pragma unroll
        for (int u = 0; u < FGS_V; u++) {
            float tt;
            fgs_step(w[u], c[u], tt, lambda, coef_prev, prev_cur, prev_tmp);
            tmpI[(size_t)(t * FGS_V + u) * rw] = tt;
            curI[(size_t)(t * FGS_V + u) * rw] = c[u];
        }
    };
    if (nblk > 0) {
        float cn[FGS_VD][FGS_V], wn[FGS_VD][FGS_V];
#pragma unroll
        for (int p = 0; p < FGS_VD; p++) ld_trip(curI, wvI, p, cn[p], wn[p]);
#pragma unroll
        for (int p = 0; p < FGS_VD; p++)
#pragma unroll
            for (int u = 0; u < FGS_V; u++) curI[(size_t)(p * FGS_V + u) * rw] = cn[p][u];   // unchanged
        for (int blk = 0; blk < nblk; blk++) {
#pragma unroll
            for (int p = 0; p < FGS_VD; p++) {
                const int t = blk * FGS_VD + p;
                fwd_trip(t, cn[p], wn[p]);
                ld_trip(curI, wvI, min(t + FGS_VD, ntrip - 1), cn[p], wn[p]);   // past the last block: an unused, in-bounds prefetch
            }
        }
    }
    for (int t = nblk * FGS_VD; t < ntrip; t++) {
        float c[FGS_V], w[FGS_V];
        ld_trip(curI, wvI, t, c, w);
        fwd_trip(t, c, w);
    }
    for (int i = ntrip * FGS_V; i < rh; i++) {
        float c = curI[(size_t)i * rw], tt;
        fgs_step(wvI[(size_t)i * rw], c, tt, lambda, coef_prev, prev_cur, prev_tmp);
        tmpI[(size_t)i * rw] = tt;
        curI[(size_t)i * rw] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // backward: row rh-1 is final after the forward sweep (it is stored again unchanged)
    float next = prev_cur;
    auto back_trip = [&](int t, const float (&c)[FGS_V], const float (&tt)[FGS_V]) {
#pragma unroll
        for (int u = FGS_V - 1; u >= 0; u--) {
            const int i = t * FGS_V + u;
            next = (i == rh - 1) ? c[u] : c[u] - tt[u] * next;
            curI[(size_t)i * rw] = next;
        }
    };
    for (int i = rh - 2; i >= ntrip * FGS_V; i--) {
        next = curI[(size_t)i * rw] - tmpI[(size_t)i * rw] * next;
        curI[(size_t)i * rw] = next;
    }
    for (int t = ntrip - 1; t >= nblk * FGS_VD; t--) {
        float c[FGS_V], tt[FGS_V];
        ld_trip(curI, tmpI, t, c, tt);
        back_trip(t, c, tt);
    }
    if (nblk > 0) {
        float cn[FGS_VD][FGS_V], tn[FGS_VD][FGS_V];
        const int ttop = nblk * FGS_VD - 1;
#pragma unroll
        for (int p = 0; p < FGS_VD; p++) ld_trip(curI, tmpI, ttop - p, cn[p], tn[p]);
#pragma unroll
        for (int p = 0; p < FGS_VD; p++)
#pragma unroll
            for (int u = 0; u < FGS_V; u++) curI[(size_t)((ttop - p) * FGS_V + u) * rw] = cn[p][u];   // unchanged
        for (int blk = 0; blk < nblk; blk++) {
#pragma unroll
            for (int p = 0; p < FGS_VD; p++) {
                const int t = ttop - blk * FGS_VD - p;
                back_trip(t, cn[p], tn[p]);
                ld_trip(curI, tmpI, max(t - FGS_VD, 0), cn[p], tn[p]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Rectification-map generation.  The module marks the camera as changed on
// every frame (DispModule.cpp:684 -> stereoCamera.cpp:1505-1536), so rectifyImages re-runs
// cv::initUndistortRectifyMap twice (stereoCamera.cpp:569-570) and updateMappings re-runs
// cv::undistortPoints over the whole pixel grid twice (:2462-2463) per frame on one CPU thread.
// Here both are per-pixel FP64 kernels fed by the 3x3 / 3x4 matrices; the maps never leave HBM.
// ---------------------------------------------------------------------------------------------
struct CamModel {
    double m[9];            // inv(P[:3,:3]*R) for the forward map, P[:3,:3]*R for the inverse mapper
    double fx, fy, cx, cy;  // camera matrix K
    double k[8];            // k1 k2 p1 p2 k3 k4 k5 k6
};

// Result of cv::initUndistortRectifyMap(K, Dist, R, P, size, CV_32FC1): one workgroup per
// destination row.  OpenCV walks a row with running sums (_x += ir[0], ...), whose rounding a
// closed form does not reproduce, so three lanes first lay the sums of the row down in LDS in that
// order (W dependent adds each, a few microseconds), then every lane distorts its pixels.
__global__ __launch_bounds__(256) void k_init_rectify_map(float* __restrict__ mapx, float* __restrict__ mapy, int W, CamModel c) {
    extern __shared__ double rs[];  // [3][W]
    const int i = blockIdx.x;
    if (threadIdx.x < 3) {
        const int t = threadIdx.x;
        double v = i * c.m[3 * t + 1] + c.m[3 * t + 2];
        const double inc = c.m[3 * t];
        double* o = rs + (size_t)t * W;
        for (int j = 0; j < W; j++, v += inc) o[j] = v;
    }
    __syncthreads();
    const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3], k3 = c.k[4], k4 = c.k[5], k5 = c.k[6], k6 = c.k[7];
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const double w = 1. / rs[2 * (size_t)W + j], x = rs[j] * w, y = rs[(size_t)W + j] * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
        const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
        mapx[(size_t)i * W + j] = (float)(c.fx * xd + c.cx);
        mapy[(size_t)i * W + j] = (float)(c.fy * yd + c.cy);
    }
}

// Result of cv::undistortPoints(grid, grid, K, Dist, R, P) on the pixel grid (updateMappings):
// MapperL / MapperR, CV_32FC2.  Five fixed-point iterations of the inverse distortion per pixel.
__global__ __launch_bounds__(256) void k_undistort_grid(float2* __restrict__ mapper, int W, int H, CamModel c) {
    const int xi = blockIdx.x * blockDim.x + threadIdx.x, yi = blockIdx.y;
    if (xi >= W) return;
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = ((double)xi - c.cx) * ifx, y = ((double)yi - c.cy) * ify;
    const double x0 = x, y0 = y;
    const double* k = c.k;
#pragma unroll 1
    for (int it = 0; it < 5; it++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    const double xx = c.m[0] * x + c.m[1] * y + c.m[2], yy = c.m[3] * x + c.m[4] * y + c.m[5];
    const double ww = 1. / (c.m[6] * x + c.m[7] * y + c.m[8]);
    mapper[(size_t)yi * W + xi] = make_float2((float)(xx * ww), (float)(yy * ww));
}

}  // namespace svk
