// prototype Y: 32 machines x (NCH x 32) positions per lane, only the CURRENT 32-step chunk transposed in registers (64 VGPRs), the chunks are
// loaded + transposed again when the warm-up and the main pass need them: <= 128 VGPRs, 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vt_core.h"
typedef uint32_t u32;
#ifndef PL
#define PL 12
#endif
#ifndef NCH
#define NCH 2
#endif
#ifndef UNR
#define UNR 4
#endif
#ifndef WPE
#define WPE 4
#endif
constexpr int UNR_ = UNR;
typedef u32 v32u __attribute__((ext_vector_type(32)));

// chunk c of the lane's 32 machines: word NCH * i + c of machine i, both planes, transposed: V0[r] bit i = plane-0 bit of position r of that word
__device__ __forceinline__ void load_chunk(const uint2* __restrict__ src, int c, v32u& V0, v32u& V1) {
    // one plane at a time: 32 raw words in flight beside the resident state
    const u32* s32 = (const u32*)(src + c);
    {
        u32 a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = s32[2 * NCH * i];
        vt_transpose32(a);
#pragma unroll
        for (int i = 0; i < 32; ++i) V0[i] = a[i];
    }
    {
        u32 a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = s32[2 * NCH * i + 1];
        vt_transpose32(a);
#pragma unroll
        for (int i = 0; i < 32; ++i) V1[i] = a[i];
    }
}

template <int L>
__global__ __launch_bounds__(256, WPE) void vt_proto2_kernel(const uint2* __restrict__ in, u32* __restrict__ out, int lane_stride8) {
    __shared__ u32 cand_lds[32 * 256];
    const int tid = threadIdx.x;
    const uint2* src = in + (size_t)blockIdx.x * 256 * lane_stride8 + (size_t)tid * lane_stride8;
    VtState<L> S; vt_reset(S);
    u32 pc0 = 0, pc1 = 0, cnt = 0;
    v32u V0, V1;
    // warm-up: the last two words of the machine before (bit i <- bit i - 1)
    load_chunk(src, NCH - 2, V0, V1);
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = V0[r] + V0[r], c1 = V1[r] + V1[r]; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
    load_chunk(src, NCH - 1, V0, V1);
#pragma unroll UNR_
    for (int r = 0; r < 8; ++r) { const u32 c0 = V0[r] + V0[r], c1 = V1[r] + V1[r]; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll UNR_
    for (int r = 8; r < 32; ++r) { const u32 c0 = V0[r] + V0[r], c1 = V1[r] + V1[r]; vt_step<L, 1>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        load_chunk(src, c, V0, V1);
#pragma unroll UNR_
        for (int r = 0; r < 32; ++r) { const u32 c0 = V0[r], c1 = V1[r]; const u32 cd = vt_step<L, 2>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; cand_lds[r * 256 + tid] = cd; }
        // (stand-in for the per-wave candidate pass: the wave reads its words back)
        for (int r = 0; r < 32; ++r) cnt += __popc(cand_lds[r * 256 + (tid ^ 1)]);
    }
    out[(size_t)blockIdx.x * 256 + tid] = cnt;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int lane_stride8 = 31 * NCH;                  // lane owns machines 1..31
    const size_t n8 = (size_t)wgs * 256 * lane_stride8 + 64 * NCH;
    std::vector<uint32_t> h(n8 * 2);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    uint2* d_in; u32* d_out;
    (void)hipMalloc(&d_in, n8 * 8); (void)hipMalloc(&d_out, (size_t)wgs * 256 * 4);
    (void)hipMemcpy(d_in, h.data(), n8 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((vt_proto2_kernel<PL>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, lane_stride8);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bases = (double)wgs * 256 * lane_stride8 * 32;
        printf("Y NCH=%d L=%d WPE=%d UNR=%d wgs=%d  %.3f ms  %.1f Gbases/s owned (%.0f cycles per owned raw word and wave at 2.4 GHz x 1024 SIMDs)\n", NCH, PL, WPE, UNR, wgs, ms, bases / ms / 1e6,
               ms * 1e-3 * 2.4e9 * 1024 / (bases / 32 / 64));
    }
    std::vector<u32> o((size_t)wgs * 256);
    (void)hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (u32 v : o) tot += v;
    printf("candidates %llu (%.4f per position)\n", tot, (double)tot / ((double)wgs * 256 * NCH * 32 * 32));
    return 0;
}
