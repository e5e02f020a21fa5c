// Issue cost and semantics of the quad-SAD instructions on gfx950 (candidates for the BT cost arithmetic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_sem(uint32_t* out) {
    // S0 = bytes 10,20,30,40,50,60,70,80 ; S1 = (25,0,0,0) ; S2 = 0
    unsigned long long s0 = 0;
    const unsigned char b[8] = {10, 20, 30, 40, 50, 60, 70, 80};
    for (int i = 0; i < 8; i++) s0 |= (unsigned long long)b[i] << (8 * i);
    unsigned long long acc = 0, q, mq;
    uint32_t s1 = 25u;  // (25, 0, 0, 0)
    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=v"(q) : "v"(s0), "v"(s1), "v"(acc));
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=v"(mq) : "v"(s0), "v"(s1), "v"(acc));
    uint32_t s1b = 25u | (35u << 16);  // (25, 0, 35, 0)
    unsigned long long mq2;
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=v"(mq2) : "v"(s0), "v"(s1b), "v"(acc));
    // masked on which operand?  S0 with zero bytes, S1 nonzero
    unsigned long long s0z = 0;
    const unsigned char bz[8] = {10, 0, 30, 0, 50, 0, 70, 0};
    for (int i = 0; i < 8; i++) s0z |= (unsigned long long)bz[i] << (8 * i);
    uint32_t s1c = 0x05050505u;
    unsigned long long mq3;
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=v"(mq3) : "v"(s0z), "v"(s1c), "v"(acc));
    if (threadIdx.x == 0) {
        out[0] = (uint32_t)q; out[1] = (uint32_t)(q >> 32);
        out[2] = (uint32_t)mq; out[3] = (uint32_t)(mq >> 32);
        out[4] = (uint32_t)mq2; out[5] = (uint32_t)(mq2 >> 32);
        out[6] = (uint32_t)mq3; out[7] = (uint32_t)(mq3 >> 32);
    }
}
template <int MODE>
__global__ void k_rate(uint32_t* out, int reps, long long* cyc) {
    unsigned long long v[4];
    for (int i = 0; i < 4; i++) v[i] = threadIdx.x * 2654435761ull + i;
    uint32_t b = out[threadIdx.x & 63] | 1u;
    unsigned long long s0 = ((unsigned long long)out[(threadIdx.x + 1) & 63] << 32) | b;
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)v[i & 3] + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (MODE == 0) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(s0), "v"(b));
                if (MODE == 1) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(v[i]) : "v"(s0), "v"(b));
                if (MODE == 2) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(w[i + 4]));
                if (MODE == 3) asm volatile("v_pk_sub_u16 %0, %1, %0 clamp" : "+v"(w[i]) : "v"(b));
                if (MODE == 4) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(w[i + 4]));
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 4; i++) s ^= (uint32_t)v[i] ^ (uint32_t)(v[i] >> 32) ^ w[i];
    out[64 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    uint32_t* d; long long* dc;
    (void)hipMalloc(&d, 8 << 20); (void)hipMalloc(&dc, 8);
    (void)hipMemset(d, 0x11, 8 << 20);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[8];
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("S0 = bytes 10..80, S1 = (25,0,0,0): qsad  = %u %u %u %u\n", h[0] & 0xFFFF, h[0] >> 16, h[1] & 0xFFFF, h[1] >> 16);
    printf("                                    mqsad = %u %u %u %u\n", h[2] & 0xFFFF, h[2] >> 16, h[3] & 0xFFFF, h[3] >> 16);
    printf("S1 = (25,0,35,0):                   mqsad = %u %u %u %u\n", h[4] & 0xFFFF, h[4] >> 16, h[5] & 0xFFFF, h[5] >> 16);
    printf("S0 = 10,0,30,0,50,0,70,0 S1 = 5,5,5,5: mqsad = %u %u %u %u\n", h[6] & 0xFFFF, h[6] >> 16, h[7] & 0xFFFF, h[7] >> 16);
    (void)hipMemset(d, 0x11, 8 << 20);
    const char* names[5] = {"v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_sad_u8", "v_pk_sub_u16 clamp", "v_msad_u8"};
    for (int mode = 0; mode < 5; mode++) {
        printf("%-22s", names[mode]);
        for (int threads : {256, 1024}) {
            const int reps = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int it = 0; it < 2; it++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                if (mode == 4) hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(threads), 0, 0, d, reps, dc);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long cyc = 0;
            (void)hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            const double n_instr = reps * 64.0;
            const double waves_per_simd = threads / 256.0;
            printf("  %4d thr: wave %.2f cyc/instr, SIMD %.2f cyc/instr (%.3f ms)", threads, cyc / n_instr, cyc / n_instr / waves_per_simd, ms);
        }
        printf("\n");
    }
    return 0;
}
