// Does gfx950 execute the wave-wide DPP shifts (wave_shr:1 / wave_shl:1), and at what issue cost?
// Build: hipcc -O3 --offload-arch=gfx950 -o wave_shr wave_shr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_check(uint32_t* out) {
    const uint32_t v = 1000u + threadIdx.x;
    uint32_t a, b, c;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, 7\n\ts_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(a) : "v"(v));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, 7\n\ts_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "=&v"(b) : "v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(c) : "v"(v));
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
    out[128 + threadIdx.x] = c;
}
template <int MODE>
__global__ void k_rate(uint32_t* out, int reps, long long* cyc) {
    uint32_t v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i;
    uint32_t b = out[threadIdx.x & 63];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(b));
                if (MODE == 1) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(b));
                if (MODE == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    uint32_t* d; long long* dc;
    hipMalloc(&d, 4 << 20); hipMalloc(&dc, 8);
    hipMemset(d, 0, 4 << 20);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(192);
    hipMemcpy(h.data(), d, 192 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const uint32_t ea = i == 0 ? 7u : 1000u + i - 1, eb = i == 63 ? 7u : 1000u + i + 1, ec = (i == 0 ? 0u : 1000u + i - 1) + 1000u + i;
        if (h[i] != ea || h[64 + i] != eb || h[128 + i] != ec) { if (bad < 8) printf("lane %d: shr %u (exp %u) shl %u (exp %u) add %u (exp %u)\n", i, h[i], ea, h[64 + i], eb, h[128 + i], ec); bad++; }
    }
    printf("wave_shr / wave_shl check: %s (%d bad lanes)\n", bad ? "FAIL" : "ok", bad);
    const char* names[3] = {"v_add_u32_dpp wave_shr:1", "v_add_u32_dpp row_shr:1", "v_add_u32"};
    for (int mode = 0; mode < 3; mode++) {
        printf("%-26s", names[mode]);
        for (int threads : {64, 256, 1024}) {
            const int reps = 4000;
            long long cyc = 0;
            for (int it = 0; it < 2; it++) {
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                hipDeviceSynchronize();
            }
            hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            const double per = (double)cyc / (reps * 64.0);       // cycles per instruction of one wave
            const double waves_per_simd = threads / 64.0 / 4.0;
            printf("  %4d thr: %.2f cyc/instr/wave -> %.2f cyc/instr/SIMD", threads, per, per / (waves_per_simd < 1 ? 1 : waves_per_simd));
        }
        printf("\n");
    }
    return bad ? 1 : 0;
}
