// prototype: the vertical filter's loops as a kernel, to look at the ISA (registers, instruction mix) and to time on the GPU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vt_core.h"
typedef uint32_t u32;
#ifndef PL
#define PL 12
#endif
#ifndef UNR
#define UNR 32
#endif
constexpr int UNR_ = UNR;
typedef u32 v32u __attribute__((ext_vector_type(32)));

template <int L>
__global__ __launch_bounds__(256, 2) void vt_proto_kernel(const uint4* __restrict__ in, u32* __restrict__ out, int lane_stride16) {
    __shared__ u32 cand_lds[64 * 256];
    const int tid = threadIdx.x;
    const uint4* src = in + (size_t)blockIdx.x * 256 * lane_stride16 + (size_t)tid * lane_stride16;
    u32 a0[32], a1[32], b0[32], b1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { const uint4 q = src[i]; a0[i] = q.x; a1[i] = q.y; b0[i] = q.z; b1[i] = q.w; }
    vt_transpose32(a0); vt_transpose32(a1); vt_transpose32(b0); vt_transpose32(b1);
    v32u A0, A1, B0, B1;
#pragma unroll
    for (int i = 0; i < 32; ++i) { A0[i] = a0[i]; A1[i] = a1[i]; B0[i] = b0[i]; B1[i] = b1[i]; }
    VtState<L> S; vt_reset(S);
    u32 pc0 = 0, pc1 = 0;
    // warm-up: the segment of the machine before (bit i <- bit i - 1)
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = A0[r] + A0[r], c1 = A1[r] + A1[r]; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll UNR_
    for (int r = 0; r < 8; ++r) { const u32 c0 = B0[r] + B0[r], c1 = B1[r] + B1[r]; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll UNR_
    for (int r = 8; r < 32; ++r) { const u32 c0 = B0[r] + B0[r], c1 = B1[r] + B1[r]; vt_step<L, 1>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
    u32 any = 0;
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = A0[r], c1 = A1[r]; const u32 cd = vt_step<L, 2>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; cand_lds[r * 256 + tid] = cd; any |= cd; }
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = B0[r], c1 = B1[r]; const u32 cd = vt_step<L, 2>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; cand_lds[(32 + r) * 256 + tid] = cd; any |= cd; }
    __syncthreads();
    u32 cnt = 0;
    for (int r = 0; r < 64; ++r) cnt += __popc(cand_lds[r * 256 + (tid ^ 1)]);
    out[(size_t)blockIdx.x * 256 + tid] = cnt + (any & 1u);
}


// one word per machine: 32 machines x 32 positions per lane; warm-up from the two machines before (bits i-2, i-1); lane owns machines 2..31
template <int L>
__global__ __launch_bounds__(256, 3) void vt_proto1_kernel(const uint2* __restrict__ in, u32* __restrict__ out, int lane_stride8) {
    __shared__ u32 cand_lds[32 * 256];
    const int tid = threadIdx.x;
    const uint2* src = in + (size_t)blockIdx.x * 256 * lane_stride8 + (size_t)tid * lane_stride8;
    u32 a0[32], a1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { const uint2 q = src[i]; a0[i] = q.x; a1[i] = q.y; }
    vt_transpose32(a0); vt_transpose32(a1);
    v32u A0, A1;
#pragma unroll
    for (int i = 0; i < 32; ++i) { A0[i] = a0[i]; A1[i] = a1[i]; }
    VtState<L> S; vt_reset(S);
    u32 pc0 = 0, pc1 = 0;
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = A0[r] << 2, c1 = A1[r] << 2; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll UNR_
    for (int r = 0; r < 8; ++r) { const u32 c0 = A0[r] + A0[r], c1 = A1[r] + A1[r]; vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
#pragma unroll UNR_
    for (int r = 8; r < 32; ++r) { const u32 c0 = A0[r] + A0[r], c1 = A1[r] + A1[r]; vt_step<L, 1>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; }
    u32 any = 0;
#pragma unroll UNR_
    for (int r = 0; r < 32; ++r) { const u32 c0 = A0[r], c1 = A1[r]; const u32 cd = vt_step<L, 2>(S, c0, c1, pc0, pc1, 0u); pc0 = c0; pc1 = c1; cand_lds[r * 256 + tid] = cd; any |= cd; }
    __syncthreads();
    u32 cnt = 0;
    for (int r = 0; r < 32; ++r) cnt += __popc(cand_lds[r * 256 + (tid ^ 1)]);
    out[(size_t)blockIdx.x * 256 + tid] = cnt + (any & 1u);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int lane_stride16 = 31;                       // 62 words of 8 bytes = 31 x 16 bytes per lane
    const size_t n16 = (size_t)wgs * 256 * lane_stride16 + 64;
    std::vector<uint32_t> h(n16 * 4);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    uint4* d_in; u32* d_out;
    hipMalloc(&d_in, n16 * 16); hipMalloc(&d_out, (size_t)wgs * 2 * 256 * 4);
    hipMemcpy(d_in, h.data(), n16 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((vt_proto_kernel<PL>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, lane_stride16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bases = (double)wgs * 256 * 62 * 32;
        printf("L=%d wgs=%d  %.3f ms  %.1f Gbases/s owned (%.0f cycles per owned raw word and wave at 2.4 GHz x 1024 SIMDs)\n", PL, wgs, ms, bases / ms / 1e6,
               ms * 1e-3 * 2.4e9 * 1024 / (bases / 32 / 64));
    }
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((vt_proto1_kernel<PL>), dim3(wgs * 2), dim3(256), 0, 0, (const uint2*)d_in, d_out, 30);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bases = (double)wgs * 2 * 256 * 30 * 32;
        printf("NCH=1 L=%d wgs=%d  %.3f ms  %.1f Gbases/s owned (%.0f cycles per owned raw word and wave)\n", PL, wgs * 2, ms, bases / ms / 1e6,
               ms * 1e-3 * 2.4e9 * 1024 / (bases / 32 / 64));
    }
    std::vector<u32> o((size_t)wgs * 256);
    hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (u32 v : o) tot += v;
    printf("candidates %llu (%.4f per position)\n", tot, (double)tot / ((double)wgs * 256 * 64 * 32));
    return 0;
}
