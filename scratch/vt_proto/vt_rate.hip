// what does one step of the vertical filter cost as a function of the waves per SIMD?  (registers only: no loads, no transposes; occupancy set by dynamic LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "vt_core.h"
typedef uint32_t u32;
#ifndef PL
#define PL 12
#endif
template <int L, int MODE, int LDSW>
__global__ __launch_bounds__(256) void rate_kernel(u32* __restrict__ out, u32 seed, int steps) {
    extern __shared__ u32 dyn[];
    VtState<L> S; vt_reset(S);
    u32 c0 = seed ^ (threadIdx.x * 2654435761u), c1 = seed * 40503u + threadIdx.x * 97u, pc0 = 0, pc1 = 0, any = 0;
#pragma unroll 4
    for (int r = 0; r < steps; ++r) {
        const u32 cd = vt_step<L, MODE>(S, c0, c1, pc0, pc1, 0u);
        if (LDSW) dyn[(r & 31) * 256 + threadIdx.x] = cd;
        any |= cd;
        pc0 = c0; pc1 = c1;
        c0 = __builtin_amdgcn_alignbit(c0, c1, 7) ^ c1; c1 = __builtin_amdgcn_alignbit(c1, c0, 13) + c0;
    }
    u32 acc = any;
    for (int d = 1; d <= VtGeo<L>::DMAX; ++d) acc ^= S.s0[d] ^ S.s1[d];
    if (acc == 0x12345678u) out[threadIdx.x] = acc + dyn[0];
}
template <int MODE, int LDSW> void run(const char* name, int wps, int instr_per_step) {
    u32* d; (void)hipMalloc(&d, 4096);
    const int steps = 2048;
    // occupancy: wps workgroups of 4 waves per CU = wps waves per SIMD, forced by the dynamic LDS size (160 KB per CU)
    const size_t lds = wps >= 8 ? 16384 : (size_t)(160 * 1024 / wps) - 1024;
    (void)hipFuncSetAttribute((const void*)rate_kernel<PL, MODE, LDSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wps * 4;
    hipLaunchKernelGGL((rate_kernel<PL, MODE, LDSW>), dim3(blocks), dim3(256), lds, 0, d, 1u, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<PL, MODE, LDSW>), dim3(blocks), dim3(256), lds, 0, d, 1u, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc_per_step_per_simd = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * steps);
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f cycles per wave-step per SIMD (%.2f per instruction at ~%d instructions per step)\n", name, wps, ms, cyc_per_step_per_simd, cyc_per_step_per_simd / instr_per_step, instr_per_step);
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<0, 0>("MODE0 (delay line only)", w, 42);
        run<2, 0>("MODE2 (full step)", w, 108);
        run<2, 1>("MODE2 + ds_write per step", w, 109);
    }
    return 0;
}
