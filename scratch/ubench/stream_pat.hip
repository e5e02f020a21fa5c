// HBM rate of the sweeps' access pattern against a band-skewed layout (one contiguous chunk per workgroup and step).
// 256 workgroups x 60 row streams x 256 bytes per step, one barrier per step, loads two steps ahead, like k_sweep.
//   stream_pat <layout 0|1> <mode 0 read C | 1 read C + write S | 2 read C + read S> [frames per pipeline]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
constexpr int H = 480, W1 = 512, R = 60, NB = H / R, PXW = 64 /* dwords per pixel */;
__device__ __forceinline__ size_t addr(int layout, int f, int b, int r, int x) {
    if (layout == 0) return ((size_t)(f * H + b * R + r) * W1 + x) * PXW;
    return (((size_t)(f * NB + b) * W1 + ((x + 2 * r) % W1)) * R + r) * PXW;
}
__global__ __launch_bounds__(1024) void k(const uint4* __restrict__ C, uint4* __restrict__ S, uint4* __restrict__ sink, int layout, int mode, int fpp) {
    const int tid = threadIdx.x, r = tid >> 4, q = tid & 15;
    const int b = blockIdx.x % NB, pipe = blockIdx.x / NB;
    const bool row = r < R;
    const int total = fpp * W1, steps = total + 2 * R + 2;
    uint4 c[2] = {}, s[2] = {}, acc = {};
    auto off = [&](int t) -> size_t {
        int u = t - 2 * r; u = (row && u >= 0 && u < total) ? u : 0;
        return addr(layout, pipe * fpp + u / W1, b, row ? r : 0, u % W1) / 4 + q;
    };
    for (int j = 0; j < 2; j++) { c[j] = C[off(j)]; if (mode == 2) s[j] = S[off(j)]; }
    for (int t = 0; t < steps; t += 2) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            acc.x ^= c[j].x + s[j].x; acc.y ^= c[j].y + s[j].y; acc.z += c[j].z ^ s[j].z; acc.w += c[j].w ^ s[j].w;
            const size_t o = off(t + j);
            if (mode == 1) S[o] = acc;
            const size_t n = off(t + j + 2);
            c[j] = C[n];
            if (mode == 2) s[j] = S[n];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    sink[blockIdx.x * 1024 + tid] = acc;
}
int main(int argc, char** argv) {
    const int layout = atoi(argv[1]), mode = atoi(argv[2]), fpp = argc > 3 ? atoi(argv[3]) : 16;
    const int pipes = 32;
    const size_t bytes = (size_t)pipes * fpp * H * W1 * PXW * 4;
    uint4 *C, *S, *sink;
    if (hipMalloc(&C, bytes) || hipMalloc(&S, bytes) || hipMalloc(&sink, 256 * 1024 * 16)) { printf("alloc failed\n"); return 1; }
    hipMemset(C, 1, bytes); hipMemset(S, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(pipes * NB), dim3(1024), 0, 0, C, S, sink, layout, mode, fpp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double moved = (double)bytes * (mode == 0 ? 1 : 2);
    printf("layout %d mode %d: %.2f ms, %.2f TB/s (%.1f GB)\n", layout, mode, best, moved / best / 1e9, moved / 1e9);
    return hipGetLastError() != hipSuccess;
}
