// synth.hip — synthetic HiFi-shaped reads, generated on the device (SURVEY.md §8d inputs).
//
// Counter-based and integer-only: every byte is a pure function of (seed, read index, position), so
// rust_mdbg_amd/synth.py regenerates the same reads on the CPU bit for bit (tests/test_synth.py).
//   genome[g]   = "ACGT"[rnd(seed, g, 0x47) >> 62]
//   read r      : span ~ mean + sd * IrwinHall(4) clipped, uniform start, random strand; per genome position an
//                 error with probability err_ppm/1e6, equally substitution / insertion / deletion.
#include "mdbg_dev.h"

__host__ __device__ inline u64 sm64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline u64 rnd3(u64 seed, u64 a, u64 b) { return sm64(sm64(seed ^ a) + b); }

struct SynthP { u64 seed, genome_len, first_read; u32 mean_len, sd_len, min_len, max_len, thr24; };

__host__ __device__ inline void read_geom(const SynthP& P, u64 r, u64& start, u32& span, u32& strand) {
    const u64 v = rnd3(P.seed + 1, r, 1);
    const int64_t s = (int64_t)((v & 0xFFFF) + ((v >> 16) & 0xFFFF) + ((v >> 32) & 0xFFFF) + ((v >> 48) & 0xFFFF));
    int64_t len = (int64_t)P.mean_len + (s * (int64_t)P.sd_len) / 37837 - (131070 * (int64_t)P.sd_len) / 37837;
    if (len < (int64_t)P.min_len) len = P.min_len;
    if (len > (int64_t)P.max_len) len = P.max_len;
    if ((u64)len > P.genome_len) len = (int64_t)P.genome_len;
    span = (u32)len;
    start = rnd3(P.seed + 1, r, 2) % (P.genome_len - (u64)len + 1);
    strand = (u32)(rnd3(P.seed + 1, r, 3) & 1);
}
// bases emitted for genome offset i of read r: n in {0,1,2}, codes c0,c1 (0..3 = A,C,G,T)
__host__ __device__ inline u32 emit_at(const SynthP& P, u64 r, u64 start, u32 i, u32& c0, u32& c1) {
    const u32 g = (u32)(rnd3(P.seed, start + i, 0x47) >> 62);
    const u64 e = rnd3(P.seed + 2, r, i);
    c0 = g; c1 = 0;
    if ((u32)(e >> 40) < P.thr24) {
        const u32 ty = (u32)((e >> 8) % 3);
        if (ty == 0) { c0 = (g + 1 + (u32)((e >> 4) % 3)) & 3; return 1; }
        if (ty == 1) { c1 = (u32)(e & 3); return 2; }
        return 0;
    }
    return 1;
}

__global__ __launch_bounds__(256) void synth_len_kernel(SynthP P, u64 n_reads, u64* __restrict__ lens) {
    __shared__ u32 ws[4];
    const u64 r = P.first_read + blockIdx.x;
    u64 start; u32 span, strand; read_geom(P, r, start, span, strand);
    u32 n = 0, c0, c1;
    for (u32 i = threadIdx.x; i < span; i += 256) n += emit_at(P, r, start, i, c0, c1);
    for (int d = 32; d; d >>= 1) n += __shfl_down(n, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) lens[blockIdx.x] = (u64)ws[0] + ws[1] + ws[2] + ws[3];
    (void)n_reads;
}

// exclusive scan of u64 lens -> offsets[n+1], single workgroup
__global__ __launch_bounds__(1024) void scan_u64_kernel(u64 n, const u64* __restrict__ in, u64* __restrict__ out) {
    __shared__ u64 ws[16]; __shared__ u64 run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (u64 i0 = 0; i0 < n; i0 += 1024) {
        const u64 i = i0 + tid;
        const u64 v = i < n ? in[i] : 0;
        u64 inc = v;
        for (int d = 1; d < 64; d <<= 1) { u64 t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        u64 b = run, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) b += ws[q]; tot += ws[q]; }
        if (i < n) out[i] = b + inc - v;
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) out[n] = run;
}

__global__ __launch_bounds__(256) void synth_write_kernel(SynthP P, const u64* __restrict__ offsets, u8* __restrict__ bases) {
    __shared__ u32 tmp[8];
    const u64 r = P.first_read + blockIdx.x;
    u64 start; u32 span, strand; read_geom(P, r, start, span, strand);
    const u64 o0 = offsets[blockIdx.x], len = offsets[blockIdx.x + 1] - o0;
    u32 running = 0;
    for (u32 i0 = 0; i0 < span; i0 += 256) {
        const u32 i = i0 + threadIdx.x;
        u32 c0 = 0, c1 = 0, n = 0;
        if (i < span) n = emit_at(P, r, start, i, c0, c1);
        u32 total;
        const u32 rank = block_excl_scan_256(n, tmp, total);
        const u32 cc[2] = {c0, c1};
        for (u32 q = 0; q < n; ++q) {
            const u64 f = running + rank + q;                    // index in forward orientation
            if (!strand) bases[o0 + f] = "ACGT"[cc[q]];
            else bases[o0 + (len - 1 - f)] = "ACGT"[3 - cc[q]];  // reverse complement
        }
        running += total;
    }
}

void launch_synth(const SynthP& P, u64 n_reads, u64* lens, u64* offsets, u8* bases, int phase, hipStream_t s) {
    if (phase == 0) {
        hipLaunchKernelGGL(synth_len_kernel, dim3((unsigned)n_reads), dim3(256), 0, s, P, n_reads, lens);
        hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, s, n_reads, lens, offsets);
    } else {
        hipLaunchKernelGGL(synth_write_kernel, dim3((unsigned)n_reads), dim3(256), 0, s, P, offsets, bases);
    }
}
