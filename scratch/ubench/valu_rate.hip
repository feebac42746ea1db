// Micro-benchmark: issue rate of integer VALU instructions on gfx950 (cycles per wave64 instruction per SIMD).
// Each thread runs NCHAIN independent dependency chains of the instruction under test; 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <cstdlib>
#include <cstring>

#ifndef NCHAIN
#define NCHAIN 8
#endif
#define UNROLL 32

#define DEFKERNEL(NAME, ASM)                                                                          \
__global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed, int iters) {           \
    uint32_t a[NCHAIN];                                                                               \
    uint32_t b = seed ^ threadIdx.x, c = seed * 3u + 1u;                                               \
    for (int i = 0; i < NCHAIN; ++i) a[i] = seed + i * 77u + threadIdx.x;                              \
    for (int it = 0; it < iters; ++it) {                                                              \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                          \
            _Pragma("unroll") for (int i = 0; i < NCHAIN; ++i) { ASM; }                               \
        }                                                                                             \
    }                                                                                                 \
    uint32_t r = 0;                                                                                   \
    for (int i = 0; i < NCHAIN; ++i) r ^= a[i];                                                       \
    if (r == 0x12345678u) out[threadIdx.x] = r;                                                       \
}

DEFKERNEL(xor,      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(and,      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(add,      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(lshl1,    asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i])))
DEFKERNEL(lshr1,    asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i])))
DEFKERNEL(lshlv,    asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(bfe,      asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i])))
DEFKERNEL(lshl_or,  asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(and_or,   asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(bitop3,   asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(min,      asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(cndmask,  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : ))
DEFKERNEL(cndmask64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : ))
DEFKERNEL(cmp_vcc,  asm volatile("v_cmp_ge_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(cmp_e64,  asm volatile("v_cmp_ge_u32_e64 s[20:21], %0, %1\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21", "s22", "s23"))
DEFKERNEL(cmp_only, asm volatile("v_cmp_ge_u32_e64 s[20:21], %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "s20", "s21"))
DEFKERNEL(mul_lo,   asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(mad24,    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(perm,     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(dot4,     asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(fma,      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(pk_add16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(pk_lshl16, asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(a[i])))
DEFKERNEL(sdwa_or,  asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(mov_dpp,  asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)))

DEFKERNEL(or_,      asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(sub,      asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(lshrv,    asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c)))
DEFKERNEL(ashr,     asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i])))
DEFKERNEL(bfe_i,    asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[i])))
DEFKERNEL(and_lit,  asm volatile("v_and_b32 %0, 0x12345678, %0" : "+v"(a[i])))
DEFKERNEL(and_sgpr, asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)))
DEFKERNEL(bitop3_s, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xea" : "+v"(a[i]) : "s"(seed), "v"(c)))
DEFKERNEL(bitop3_l, asm volatile("v_bitop3_b32 %0, %0, 32, %1 bitop3:0xea" : "+v"(a[i]) : "v"(c)))
DEFKERNEL(add3,     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(or3,      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(xad,      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(add_lshl, asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(not_,     asm volatile("v_not_b32 %0, %0" : "+v"(a[i])))
DEFKERNEL(mov,      asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(bfrev,    asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i])))
DEFKERNEL(max,      asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(cmp32_xor, asm volatile("v_cmp_ge_u32 vcc, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(cmp32_cnd, asm volatile("v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(addc32,   asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(add_co,   asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(sub_co,   asm volatile("v_sub_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc"))
DEFKERNEL(lshl64,   asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(*(unsigned long long*)&a[i & ~1])))
DEFKERNEL(sad,      asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(mul24,    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(popc,     asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(mbcnt,    asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEFKERNEL(fmac,     asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEFKERNEL(pk_fma,   asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(unsigned long long*)&a[i & ~1]) : "v"(*(unsigned long long*)&b)))

struct K { const char* name; void (*fn)(uint32_t*, uint32_t, int); int per; };
int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    K ks[] = { {"v_xor_b32", k_xor, 1}, {"v_and_b32", k_and, 1}, {"v_add_u32", k_add, 1}, {"v_lshlrev_b32 const", k_lshl1, 1}, {"v_lshrrev_b32 const", k_lshr1, 1},
               {"v_lshlrev_b32 var", k_lshlv, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1},
               {"v_bitop3_b32", k_bitop3, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_min_u32", k_min, 1}, {"v_cndmask_b32 vcc", k_cndmask, 1}, {"v_cndmask_b32_e64 sgpr", k_cndmask64, 1},
               {"v_cmp(vcc)+v_addc(vcc) pair", k_cmp_vcc, 2}, {"v_cmp_e64(sgpr)+v_addc_e64 pair", k_cmp_e64, 2}, {"v_cmp_e64+v_xor pair", k_cmp_only, 2},
               {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_perm_b32", k_perm, 1}, {"v_dot4_u32_u8", k_dot4, 1},
               {"v_fma_f32", k_fma, 1}, {"v_pk_add_u16", k_pk_add16, 1}, {"v_pk_lshlrev_b16", k_pk_lshl16, 1}, {"v_or_b32_sdwa", k_sdwa_or, 1}, {"v_xor_b32_dpp", k_mov_dpp, 1},
               {"v_or_b32", k_or_, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshrrev_b32 var", k_lshrv, 1}, {"v_ashrrev_i32 const", k_ashr, 1}, {"v_bfe_i32", k_bfe_i, 1},
               {"v_and_b32 literal", k_and_lit, 1}, {"v_and_b32 sgpr", k_and_sgpr, 1}, {"v_bitop3 sgpr operand", k_bitop3_s, 1}, {"v_bitop3 inline const", k_bitop3_l, 1},
               {"v_add3_u32", k_add3, 1}, {"v_or3_b32", k_or3, 1}, {"v_xad_u32", k_xad, 1}, {"v_add_lshl_u32", k_add_lshl, 1}, {"v_not_b32", k_not_, 1}, {"v_mov_b32", k_mov, 1},
               {"v_bfrev_b32", k_bfrev, 1}, {"v_max_u32", k_max, 1}, {"v_cmp_e32(vcc)+v_xor pair", k_cmp32_xor, 2}, {"v_cmp_e32+v_cndmask_e32 pair", k_cmp32_cnd, 2},
               {"v_addc_co_u32 e32", k_addc32, 1}, {"v_add_co_u32", k_add_co, 1}, {"v_sub_co_u32", k_sub_co, 1}, {"v_lshlrev_b64", k_lshl64, 1}, {"v_sad_u32", k_sad, 1},
               {"v_mul_u32_u24", k_mul24, 1}, {"v_bcnt_u32_b32", k_popc, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_pk_fma_f32", k_pk_fma, 1} };
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;   // Hz
    const int wpc = getenv("WGS") ? atoi(getenv("WGS")) : 8; const int blocks = cus * wpc;            // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d clock %.0f MHz\n", cus, clk / 1e6);
    const char* only = getenv("ONLY");
    for (auto& k : ks) {
        if (only && !strstr(only, k.name)) continue;
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1u, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr_per_simd = (double)wpc * iters * UNROLL * NCHAIN * k.per;     // 8 waves per SIMD
        const double cyc = ms * 1e-3 * clk / wave_instr_per_simd;
        printf("%-34s %8.3f ms  %.2f cycles per wave64 instruction per SIMD\n", k.name, ms, cyc);
    }
    return 0;
}
