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
DEF_KERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEF_KERNEL(k_cndvcc0, "s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
DEF_KERNEL(k_cmp64, "v_cmp_lt_u32_e64 s[10:11], %0, %1")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1")
DEF_KERNEL(k_pksubi, "v_pk_sub_i16 %0, %0, %1")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
DEF_KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %1")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_cnd64vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
DEF_KERNEL(k_cndsw, "v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL(k_cmpcnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_cmpcnd64, "v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEF_KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_KERNEL(k_or, "v_or_b32 %0, %0, %1")
DEF_KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
DEF_KERNEL(k_subsat, "v_sub_u16 %0, %0, %1 clamp")
#define RUN(K, ILP, name)                                                                      \
    {                                                                                          \
        printf("%-22s ILP=%d:", name, ILP);                                                    \
        for (int cfg : {64, 256, 1024, 2048}) {                                     \
            const int reps = 4000, threads = cfg > 1024 ? 1024 : cfg, grid = 256 * (cfg > 1024 ? cfg / 1024 : 1); \
            hipLaunchKernelGGL((K<ILP>), dim3(grid), dim3(threads), 0, 0, d, reps, dc);        \
            hipDeviceSynchronize();                                                            \
            hipEventRecord(e0, 0);                                                             \
            hipLaunchKernelGGL((K<ILP>), dim3(grid), dim3(threads), 0, 0, d, reps, dc);        \
            hipEventRecord(e1, 0); hipEventSynchronize(e1);                                    \
            float ms; hipEventElapsedTime(&ms, e0, e1);                                        \
            long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);                          \
            const double pw = (double)c / (reps * 64.0);                                       \
            const double wps = (double)grid * ((threads + 63) / 64) / 1024.0;                  \
            const double ns_simd = ms * 1e6 / (wps * reps * 64.0);                             \
            printf("  %4.1f w/SIMD: %.2f tick/instr/wave, %.3f ns per SIMD issue (%.2f cyc @2.4GHz)", wps, pw, ns_simd, ns_simd * 2.4); \
        }                                                                                      \
        printf("\n");                                                                          \
    }
int main() {
    uint32_t* d; long long* dc;
    hipMalloc(&d, 1024 * 1024 * 4); hipMemset(d, 1, 1024 * 1024 * 4); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipMalloc(&dc, 8);
    RUN(k_pkmin, 8, "v_pk_min_u16") RUN(k_pkadd, 8, "v_pk_add_u16") RUN(k_pksubs, 8, "v_pk_sub_u16 clamp")
    RUN(k_min32, 8, "v_min_u32") RUN(k_align, 8, "v_alignbit_b32") RUN(k_mindpp, 8, "v_min_u32_dpp")
    RUN(k_movdpp, 8, "v_mov_b32_dpp") RUN(k_add32, 8, "v_add_u32") RUN(k_cndmask, 8, "v_cndmask_b32") RUN(k_mullo, 8, "v_mul_lo_u32")
    RUN(k_cnd64, 8, "v_cndmask_e64 sgpr") RUN(k_cndvcc0, 8, "s_mov vcc + v_cndmask") RUN(k_cmp, 8, "v_cmp_lt_u32 vcc") RUN(k_cmp64, 8, "v_cmp_lt_u32 e64")
    RUN(k_perm, 8, "v_perm_b32") RUN(k_add3, 8, "v_add3_u32") RUN(k_pksubi, 8, "v_pk_sub_i16") RUN(k_bfi, 8, "v_bfi_b32") RUN(k_andor, 8, "v_and_or_b32")
    RUN(k_cnd64vcc, 8, "v_cndmask_e64 vcc") RUN(k_cndsw, 8, "v_cndmask vcc swapped") RUN(k_cmpcnd, 8, "cmp+cndmask vcc (2 instr)") RUN(k_cmpcnd64, 8, "cmp+cndmask sgpr (2 instr)") RUN(k_addc, 8, "v_addc_co_u32 vcc")
    RUN(k_mov, 8, "v_mov_b32") RUN(k_or, 8, "v_or_b32") RUN(k_lshl, 8, "v_lshlrev_b32") RUN(k_subsat, 8, "v_sub_u16 clamp")
    RUN(k_pkmin, 1, "v_pk_min_u16 dependent") RUN(k_mindpp, 1, "v_min_u32_dpp dependent") RUN(k_add32, 1, "v_add_u32 dependent")
    return 0;
}
