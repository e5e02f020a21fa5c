// Micro-benchmark: issue cost of the VALU instructions the SGM kernels are made of (gfx950).
// Inline asm (nothing for the compiler to fold): REP x 64 instructions per wave over ILP independent
// registers; 256 blocks (one per CU) of 64..1024 threads = 1..16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DEF_KERNEL(NAME, ASM)                                                                  \
    template <int ILP> __global__ void NAME(uint32_t* out, int reps, long long* cyc) {         \
        uint32_t v[ILP];                                                                       \
        for (int i = 0; i < ILP; i++) v[i] = threadIdx.x * 2654435761u + i;                    \
        uint32_t b = out[threadIdx.x & 63];                                                    \
        long long t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int r = 0; r < reps; r++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 64 / ILP; u++)                               \
                _Pragma("unroll") for (int i = 0; i < ILP; i++) asm volatile(ASM : "+v"(v[i]) : "v"(b)); \
        }                                                                                      \
        long long t1 = __builtin_amdgcn_s_memtime();                                           \
        uint32_t s = 0;                                                                        \
        for (int i = 0; i < ILP; i++) s ^= v[i];                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                               \
    }
DEF_KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
DEF_KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(k_pksubs, "v_pk_sub_u16 %0, %1, %0 clamp")
DEF_KERNEL(k_min32, "v_min_u32 %0, %0, %1")
DEF_KERNEL(k_align, "v_alignbit_b32 %0, %0, %1, 16")
DEF_KERNEL(k_mindpp, "v_min_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_movdpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_add32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
#define RUN(K, ILP, name)                                                                      \
    {                                                                                          \
        printf("%-22s ILP=%d:", name, ILP);                                                    \
        for (int threads : {64, 256, 512, 1024}) {                                             \
            const int reps = 2000;                                                             \
            hipLaunchKernelGGL((K<ILP>), dim3(256), dim3(threads), 0, 0, d, reps, dc);         \
            hipDeviceSynchronize();                                                            \
            long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);                          \
            const double pw = (double)c / (reps * 64.0);                                       \
            printf("  %2d waves/SIMD: %.2f cyc/instr/wave = %.2f cyc per SIMD issue", (threads + 255) / 256, pw, pw / ((threads + 255) / 256)); \
        }                                                                                      \
        printf("\n");                                                                          \
    }
int main() {
    uint32_t* d; long long* dc;
    hipMalloc(&d, 256 * 1024 * 4); hipMemset(d, 1, 256 * 1024 * 4); hipMalloc(&dc, 8);
    RUN(k_pkmin, 8, "v_pk_min_u16") RUN(k_pkadd, 8, "v_pk_add_u16") RUN(k_pksubs, 8, "v_pk_sub_u16 clamp")
    RUN(k_min32, 8, "v_min_u32") RUN(k_align, 8, "v_alignbit_b32") RUN(k_mindpp, 8, "v_min_u32_dpp")
    RUN(k_movdpp, 8, "v_mov_b32_dpp") RUN(k_add32, 8, "v_add_u32") RUN(k_cndmask, 8, "v_cndmask_b32") RUN(k_mullo, 8, "v_mul_lo_u32")
    RUN(k_pkmin, 1, "v_pk_min_u16 dependent") RUN(k_mindpp, 1, "v_min_u32_dpp dependent") RUN(k_add32, 1, "v_add_u32 dependent")
    return 0;
}
