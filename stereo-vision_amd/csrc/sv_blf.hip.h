// sv_blf.hip.h — filterBLF, BLF_ORIGINAL branch (StereoMatcher.cpp:90-99): the bilateral-grid filter
// cv_extend::bilateralFilter of modules/DisparityModule/fastBilateral.hpp:92-189 on the 16-bit
// disparity map, in FP64 like the reference, so that the result is bit-exact against the CPU:
//   * the splat adds integers (disparity values, counts) into doubles: exact, so the order in which
//     the atomics land does not matter;
//   * the six [1 2 1]/4 passes only add and halve such values: still exact (< 2^53 by far);
//   * the read-back divides sum by count per cell and blends eight cells in the order the reference
//     writes the expression (built with -ffp-contract=off: no fused multiply-add).
// Grid layout: [gh][gw][gd] cells of (sum, count) as double2, z (the value axis) fastest, as
// cv::Mat(3, {h, w, d}, CV_64FC2) lays it out (fastBilateral.hpp:126-127).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svk {

struct BlfGrid {
    int gh, gw, gd;          // cells per axis, padding included (fastBilateral.hpp:122-124)
    double vmin;             // smallest input value (:120)
    double sigma_color, sigma_space;
};
constexpr int BLF_PAD = 2;   // padding_xy = padding_z = 2 (:118)

// mm[0] = min, mm[1] = max over the map (cv::minMaxLoc, :120); mm preset to {INT_MAX, INT_MIN}
__global__ __launch_bounds__(256) void k_blf_minmax(const int16_t* __restrict__ d, size_t n, int* __restrict__ mm) {
    int lo = 32767, hi = -32768;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = d[i];
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        lo = min(lo, __shfl_xor(lo, s));
        hi = max(hi, __shfl_xor(hi, s));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm, lo);
        atomicMax(mm + 1, hi);
    }
}

// :131-143 every pixel goes into its nearest cell
__global__ __launch_bounds__(256) void k_blf_splat(const int16_t* __restrict__ d, int W, int H, BlfGrid g, double2* __restrict__ grid) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const double v = (double)d[(size_t)y * W + x];
    const size_t sx = (size_t)((double)x / g.sigma_space + 0.5) + BLF_PAD;
    const size_t sy = (size_t)((double)y / g.sigma_space + 0.5) + BLF_PAD;
    const size_t sz = (size_t)((v - g.vmin) / g.sigma_color + 0.5) + BLF_PAD;
    double2* cell = grid + (sy * g.gw + sx) * g.gd + sz;
    unsafeAtomicAdd(&cell->x, v);
    unsafeAtomicAdd(&cell->y, 1.0);
}

// :153-170 one [1 2 1]/4 pass along the axis whose element stride is `off`, interior cells only
__global__ __launch_bounds__(256) void k_blf_blur(const double2* __restrict__ src, double2* __restrict__ dst, BlfGrid g, long off) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int x = blockIdx.y + 1, y = blockIdx.z + 1;
    if (z >= g.gd - 1) return;
    const size_t c = ((size_t)y * g.gw + x) * g.gd + z;
    const double2 p = src[c - off], q = src[c + off], m = src[c];
    double2 r;
    r.x = (p.x + q.x + 2.0 * m.x) / 4.0;
    r.y = (p.y + q.y + 2.0 * m.y) / 4.0;
    dst[c] = r;
}

// :174-187 + :108-109: normalise the eight surrounding cells, blend, round half to even, saturate
__global__ __launch_bounds__(256) void k_blf_slice(const int16_t* d, int W, int H, BlfGrid g, const double2* __restrict__ grid,
                                                   int16_t* out /* may be d */) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const double z = (double)d[(size_t)y * W + x] - g.vmin;
    const double px = (double)x / g.sigma_space + (double)BLF_PAD;
    const double py = (double)y / g.sigma_space + (double)BLF_PAD;
    const double pz = z / g.sigma_color + (double)BLF_PAD;
    const size_t y0 = min((size_t)py, (size_t)g.gh - 1), y1 = min(y0 + 1, (size_t)g.gh - 1);
    const size_t x0 = min((size_t)px, (size_t)g.gw - 1), x1 = min(x0 + 1, (size_t)g.gw - 1);
    const size_t z0 = min((size_t)pz, (size_t)g.gd - 1), z1 = min(z0 + 1, (size_t)g.gd - 1);
    const double ya = py - (double)y0, xa = px - (double)x0, za = pz - (double)z0;
    auto G = [&](size_t yy, size_t xx, size_t zz) -> double {
        const double2 c = grid[(yy * g.gw + xx) * g.gd + zz];
        return c.x / (c.y != 0 ? c.y : 1.0);
    };
    const double r = (1.0 - ya) * (1.0 - xa) * (1.0 - za) * G(y0, x0, z0) + (1.0 - ya) * xa * (1.0 - za) * G(y0, x1, z0) +
                     ya * (1.0 - xa) * (1.0 - za) * G(y1, x0, z0) + ya * xa * (1.0 - za) * G(y1, x1, z0) +
                     (1.0 - ya) * (1.0 - xa) * za * G(y0, x0, z1) + (1.0 - ya) * xa * za * G(y0, x1, z1) +
                     ya * (1.0 - xa) * za * G(y1, x0, z1) + ya * xa * za * G(y1, x1, z1);
    const double q = rint(r);
    out[(size_t)y * W + x] = (int16_t)(q < -32768.0 ? -32768.0 : (q > 32767.0 ? 32767.0 : q));
}

// ---------------------------------------------------------------------------------------------
// filterBLF, BLF_CUDA branch (StereoMatcher.cpp:101-116): the joint bilateral disparity refinement
// of cv::cuda::DisparityBilateralFilter (EXT item; algorithm and the fixed read-before-write
// schedule: include/sv_hip.h, sv_filter_blf_joint_*).  One half-pass: thread = one pixel of the pass's
// checkerboard colour; `cur` is the map before the half-pass, `out` starts as a copy of it.
// tc[256] / ts[(radius+1)^2]: colour and space weight tables made on the host.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blf_joint(const int16_t* __restrict__ cur, int16_t* __restrict__ out,
                                                    const uint8_t* __restrict__ guide, const float* __restrict__ tc,
                                                    const float* __restrict__ ts, int W, int H, int t, int radius, int edge_disc,
                                                    int max_disc) {
    const int y = blockIdx.y;
    const int x = ((blockIdx.x * blockDim.x + threadIdx.x) << 1) + ((y + t) & 1);
    if (!(y > 0 && y < H - 1 && x > 0 && x < W - 1)) return;
    int dp[5];
    dp[0] = cur[(size_t)y * W + x];
    dp[1] = cur[(size_t)(y - 1) * W + x];
    dp[2] = cur[(size_t)y * W + x - 1];
    dp[3] = cur[(size_t)(y + 1) * W + x];
    dp[4] = cur[(size_t)y * W + x + 1];
    if (!(abs(dp[1] - dp[0]) >= edge_disc || abs(dp[2] - dp[0]) >= edge_disc || abs(dp[3] - dp[0]) >= edge_disc ||
          abs(dp[4] - dp[0]) >= edge_disc))
        return;
    const int ymin = max(0, y - radius), xmin = max(0, x - radius);
    const int ymax = min(H - 1, y + radius), xmax = min(W - 1, x + radius);
    float cost[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const int ic = guide[(size_t)y * W + x];
    for (int yi = ymin; yi <= ymax; yi++)
        for (int xi = xmin; xi <= xmax; xi++) {
            const float weight = tc[abs((int)guide[(size_t)yi * W + xi] - ic)] * ts[abs(y - yi) * (radius + 1) + abs(x - xi)];
            const int dr = cur[(size_t)yi * W + xi];
#pragma unroll
            for (int k = 0; k < 5; k++) cost[k] += (float)min(max_disc, abs(dr - dp[k])) * weight;
        }
    float minimum = 3.402823466e+38f;
    int best = dp[0];
#pragma unroll
    for (int k = 0; k < 5; k++)
        if (cost[k] < minimum) { minimum = cost[k]; best = dp[k]; }
    out[(size_t)y * W + x] = (int16_t)best;
}

}  // namespace svk
