// Follow-up to valu_rate.hip: when is v_cndmask_b32 (e32, mask in VCC) slow on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CND8 "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n" \
             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define CND8E "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n" \
             "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n"
#define DEF(NAME, PRE, BODY)                                                                    \
    __global__ void NAME(uint32_t* out, int reps) {                                            \
        uint32_t v[8];                                                                         \
        for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i;                      \
        uint32_t b = out[threadIdx.x & 63];                                                    \
        for (int r = 0; r < reps; r++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                      \
                asm volatile(PRE BODY : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(b) : "vcc", "s10", "s11"); \
        }                                                                                      \
        uint32_t s = 0;                                                                        \
        for (int i = 0; i < 8; i++) s ^= v[i];                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
    }
DEF(k_none, "", CND8)
DEF(k_cmp1, "v_cmp_lt_u32 vcc, %0, %8\n", CND8)
DEF(k_cmp1e, "v_cmp_lt_u32 vcc, %0, %8\n", CND8E)
DEF(k_cmp_sand, "v_cmp_lt_u32 vcc, %0, %8\n s_and_b64 vcc, vcc, exec\n", CND8)
DEF(k_cmp_sand_e, "v_cmp_lt_u32 vcc, %0, %8\n s_and_b64 vcc, vcc, exec\n", CND8E)
DEF(k_cmps_mov, "v_cmp_lt_u32_e64 s[10:11], %0, %8\n s_mov_b64 vcc, s[10:11]\n", CND8)
DEF(k_cmp_other, "v_cmp_lt_u32 vcc, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n", CND8)
#define RUN(K, n, name)                                                                        \
    {                                                                                          \
        const int reps = 4000, grid = 512, threads = 1024;                                     \
        hipLaunchKernelGGL(K, dim3(grid), dim3(threads), 0, 0, d, reps);                       \
        hipDeviceSynchronize(); hipEventRecord(e0, 0);                                         \
        hipLaunchKernelGGL(K, dim3(grid), dim3(threads), 0, 0, d, reps);                       \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                        \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                            \
        const double wps = (double)grid * 16 / 1024.0;                                         \
        printf("%-44s %6.1f cycles per block of %d instr @2.4GHz, 8 waves/SIMD\n", name, ms * 1e6 / (wps * reps * 8.0) * 2.4, n); \
    }
int main() {
    uint32_t* d; (void)hipMalloc(&d, 1024 * 1024 * 4); (void)hipMemset(d, 1, 1024 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    RUN(k_none, 8, "8 cndmask e32, vcc never written")
    RUN(k_cmp1, 9, "v_cmp vcc + 8 cndmask e32")
    RUN(k_cmp1e, 9, "v_cmp vcc + 8 cndmask e64(vcc)")
    RUN(k_cmp_sand, 10, "v_cmp + s_and vcc + 8 cndmask e32")
    RUN(k_cmp_sand_e, 10, "v_cmp + s_and vcc + 8 cndmask e64(vcc)")
    RUN(k_cmps_mov, 10, "v_cmp sgpr + s_mov vcc + 8 cndmask e32")
    RUN(k_cmp_other, 13, "v_cmp vcc + 4 v_add + 8 cndmask e32")
    return 0;
}
