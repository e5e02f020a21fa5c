// v_pk_minimum3_f16 (new on gfx950) as a packed 3-input u16 minimum: non-negative finite f16 bit patterns (0 .. 0x7BFF) order like
// the integers they spell, denormals (below 0x0400) included as long as the kernel keeps the default f16 denormal mode.
// (1) every pair (a, b) in [0, 0x7C00)^2 with a hashed third operand, both halves, against the integer minimum;
// (2) issue rate against v_pk_min_u16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d; asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__global__ void k_check(unsigned long long* bad, uint32_t lim) {
    const uint32_t a = blockIdx.x;   // 0 .. lim-1
    unsigned long long n = 0;
    for (uint32_t b = threadIdx.x; b < lim; b += blockDim.x) {
        const uint32_t c = ((a * 2654435761u) ^ (b * 40503u) ^ (b >> 3)) % lim, c2 = (c * 7u + a) % lim;
        const uint32_t got = pkmin3(a | (b << 16), b | (c << 16), c | (c2 << 16));
        const uint32_t lo = min(min(a, b), c), hi = min(min(b, c), c2);
        n += (got != (lo | (hi << 16)));
    }
    if (n) atomicAdd(bad, n);
}
template <int MODE> __global__ void k_rate(uint32_t* out, int reps) {
    uint32_t v[8]; for (int i = 0; i < 8; i++) v[i] = (threadIdx.x * 2654435761u + i) & 0x3FFF3FFFu;
    const uint32_t b = out[threadIdx.x & 63] & 0x3FFF3FFFu;
    for (int r = 0; r < reps; r++)
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
                else asm volatile("v_pk_minimum3_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
            }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    unsigned long long* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(0x7C00), dim3(256), 0, 0, bad, 0x7C00u);
    unsigned long long h = 1; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("check: %llu mismatches over %u x %u pairs x 2 halves\n", h, 0x7C00u, 0x7C00u);
    uint32_t* d; hipMalloc(&d, 1024 * 1024 * 4); hipMemset(d, 1, 1024 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(1024), 0, 0, d, 4000);
            else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(1024), 0, 0, d, 4000);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.3f ns per SIMD issue (%.2f cycles at 2.4 GHz), 4 waves per SIMD\n", mode ? "v_pk_minimum3_f16" : "v_pk_min_u16     ",
               ms * 1e6 / (4.0 * 4000 * 64), ms * 1e6 / (4.0 * 4000 * 64) * 2.4);
    }
    return h != 0;
}
