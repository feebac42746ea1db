// edges.hip — graph edges of the k-min-mer node table on the GPU (gfx950).
//
// Replaces the single-threaded edge loop of rust-mdbg (src/main.rs:1017-1117): km_index from every node's normalized
// (k-1)-prefix and (k-1)-suffix to the nodes listing it, the four orientation tests (:1062-1075), the abundance
// presimplification (:1078-1090, and its reverse :1104-1115) and overlap = min(n1.seqlen - shift(ori1), n2.seqlen - 1).
// Output order = the order in which the host emitter (mdbg_emit.cpp) and the oracle produce the edges: n1 in node-table
// order, suffix key before prefix key, listings in (node, prefix-before-suffix) order, orientations ++, +-, -+, --.
//
//   list_kernel      one listing per (node, side): hash of the normalized (k-1)-mer, sorted by rocPRIM's stable radix sort
//   query_kernel     one thread per (node, key): equal range by binary search, exact (k-1)-mer comparison, orientation
//                    tests; pass 1 counts, pass 2 writes candidate edges at scanned offsets and records presimp removals
//   filter / scatter drop candidates whose pair (either direction) was removed, compact in order
#include <algorithm>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "edges.h"

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

namespace {

constexpr u64 EMPTY64 = ~0ull;

struct Buf {
    void* p = nullptr; size_t cap = 0;
    ~Buf() { if (p) mdbg_block_free(p, cap); }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) mdbg_block_free(p, cap);
        p = nullptr; cap = 0;
        return mdbg_block_alloc(&p, bytes + bytes / 8 + 256, &cap);
    }
    template <class T> T* as() const { return (T*)p; }
};

__device__ inline u64 fmix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// KmerVec::normalize().1 of the span p[0..km): true = the reversal is the canonical form (ties included)
__device__ inline bool span_reversed(const u64* __restrict__ p, u32 km) {
    for (u32 i = 0; i < km; ++i) { const u64 a = p[i], b = p[km - 1 - i]; if (a < b) return false; if (a > b) return true; }
    return true;
}
__device__ inline u64 span_at(const u64* __restrict__ p, u32 km, bool rev, u32 j) { return rev ? p[km - 1 - j] : p[j]; }
__device__ inline u64 span_hash(const u64* __restrict__ p, u32 km, bool rev) {
    u64 h0 = 0x9E3779B97F4A7C15ull, h1 = 0xD1B54A32D192ED03ull;
    u32 j = 0;
    for (; j + 2 <= km; j += 2) {
        h0 = (h0 ^ span_at(p, km, rev, j)) * 0xff51afd7ed558ccdull;     h0 ^= h0 >> 29;
        h1 = (h1 ^ span_at(p, km, rev, j + 1)) * 0xc4ceb9fe1a85ec53ull; h1 ^= h1 >> 31;
    }
    if (j < km) { h0 = (h0 ^ span_at(p, km, rev, j)) * 0xff51afd7ed558ccdull; h0 ^= h0 >> 29; }
    const u64 h = fmix(h0 ^ ((h1 << 23) | (h1 >> 41)));
    return h == EMPTY64 ? 0 : h;
}
__device__ inline bool span_equal(const u64* __restrict__ a, bool ra, const u64* __restrict__ b, bool rb, u32 km) {
    for (u32 j = 0; j < km; ++j) if (span_at(a, km, ra, j) != span_at(b, km, rb, j)) return false;
    return true;
}

// listing value: row << 2 | suffix << 1 | reversed; ascending value = the host emitter's push_back order inside a bucket
__global__ __launch_bounds__(256) void list_kernel(EdgeNodes nd, u64* __restrict__ lh, u32* __restrict__ lv) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * nd.n) return;
    const u64 row = t >> 1; const u32 suf = (u32)(t & 1), km = nd.k - 1;
    const u64* p = nd.keys + row * nd.k + suf;
    const bool rev = span_reversed(p, km);
    lh[t] = span_hash(p, km, rev);
    lv[t] = (u32)(row << 2) | (suf << 1) | (rev ? 1u : 0u);
}

struct QueryArgs {
    EdgeNodes nd; float presimp;
    const u64* sh; const u32* sv; u64 n_list;        // sorted listings
    u32* cnt; u16* amax; const u64* off;             // per (node, key)
    u32* s_a; u32* s_b; u32* s_ov; u8* s_o;          // candidate slots: index of n1 / n2, overlap, o1 | o2 << 1 | dropped << 7
    u64* set; u64 set_mask; unsigned long long* n_removed;
};

__device__ inline void set_insert(u64* set, u64 mask, u64 key) {
    u64 s = fmix(key) & mask;
    for (;;) {
        const u64 old = atomicCAS((unsigned long long*)&set[s], (unsigned long long)EMPTY64, (unsigned long long)key);
        if (old == EMPTY64 || old == key) return;
        s = (s + 1) & mask;
    }
}
__device__ inline bool set_has(const u64* set, u64 mask, u64 key) {
    u64 s = fmix(key) & mask;
    for (;;) {
        const u64 v = set[s];
        if (v == key) return true;
        if (v == EMPTY64) return false;
        s = (s + 1) & mask;
    }
}

template <bool WRITE>
__global__ __launch_bounds__(256) void query_kernel(QueryArgs a) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * a.nd.n) return;
    const u64 i = t >> 1; const u32 q = (u32)(t & 1), k = a.nd.k, km = k - 1;
    const u64* k1 = a.nd.keys + i * k;
    // key 0 = normalized suffix, key 1 = normalized prefix (main.rs:1051-1053)
    const u64* qp = q == 0 ? k1 + 1 : k1;
    const bool qrev = span_reversed(qp, km);
    const u64 h = span_hash(qp, km, qrev);
    u64 lo = 0, hi = a.n_list;                       // first listing with hash >= h
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (a.sh[mid] < h) lo = mid + 1; else hi = mid; }
    u32 npot = 0; u16 amax = 0;
    const u32 npot_all = WRITE ? a.cnt[t] : 0;
    const u16 a1 = a.nd.abund[i];
    u16 aref = 0;
    if (WRITE) { const u16 am = a.amax[t]; aref = am < a1 ? am : a1; }
    u64 w = WRITE ? a.off[t] : 0;
    for (u64 x = lo; x < a.n_list && a.sh[x] == h; ++x) {
        const u32 v = a.sv[x];
        const u64 row = v >> 2; const u32 suf = (v >> 1) & 1; const bool lrev = (v & 1) != 0;
        const u64* k2 = a.nd.keys + row * k;
        if (!span_equal(qp, qrev, k2 + suf, lrev, km)) continue;          // same hash, different (k-1)-mer
        // n1.suffix = k1[1..k), reversed n1's suffix = k1[0..k-1) backwards; n2.prefix = k2[0..k-1), reversed n2's prefix = k2[1..k) backwards
        const bool b[4] = { span_equal(k1 + 1, false, k2, false, km), span_equal(k1 + 1, false, k2 + 1, true, km),
                            span_equal(k1, true, k2, false, km),       span_equal(k1, true, k2 + 1, true, km) };
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (!b[o]) continue;
            ++npot;
            const u16 a2 = a.nd.abund[row];
            if (!WRITE) { amax = a2 > amax ? a2 : amax; continue; }
            const u32 o1 = (u32)o >> 1, o2 = (u32)o & 1;                   // 0 = '+', 1 = '-'
            u8 flags = (u8)(o1 | (o2 << 1));
            const u32 ia = a.nd.index[i], ib = a.nd.index[row];
            if (a.presimp > 0.0f && npot_all >= 2 && (float)a2 < __fmul_rn(a.presimp, (float)aref)) {      // main.rs:1080-1088
                flags |= 0x80;
                set_insert(a.set, a.set_mask, ((u64)ia << 32) | ib);
                atomicAdd(a.n_removed, 1ull);
            }
            const u16 shift = o1 == 0 ? a.nd.shift[2 * i] : a.nd.shift[2 * i + 1];
            const u32 ov1 = a.nd.seqlen[i] - (u32)shift, ov2 = a.nd.seqlen[row] - 1u;                       // main.rs:1091-1092 (u32 arithmetic)
            a.s_a[w] = ia; a.s_b[w] = ib; a.s_ov[w] = ov1 < ov2 ? ov1 : ov2; a.s_o[w] = flags;
            ++w;
        }
    }
    if (!WRITE) { a.cnt[t] = npot; a.amax[t] = amax; }
}

__global__ __launch_bounds__(256) void filter_kernel(u64 n_slots, const u32* __restrict__ s_a, const u32* __restrict__ s_b, const u8* __restrict__ s_o,
                                                     const u64* __restrict__ set, u64 set_mask, bool presimp, u32* __restrict__ keep) {
    const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_slots) return;
    bool k = !(s_o[x] & 0x80);
    if (k && presimp) {
        const u64 ab = ((u64)s_a[x] << 32) | s_b[x], ba = ((u64)s_b[x] << 32) | s_a[x];
        k = !(set_has(set, set_mask, ab) || set_has(set, set_mask, ba));    // main.rs:1107-1111
    }
    keep[x] = k ? 1u : 0u;
}
__global__ __launch_bounds__(256) void scatter_kernel(u64 n_slots, const u32* __restrict__ keep, const u64* __restrict__ pos, const u32* __restrict__ s_a,
                                                      const u32* __restrict__ s_b, const u32* __restrict__ s_ov, const u8* __restrict__ s_o,
                                                      u32* __restrict__ n1, u8* __restrict__ o1, u32* __restrict__ n2, u8* __restrict__ o2, u32* __restrict__ ov) {
    const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_slots || !keep[x]) return;
    const u64 d = pos[x];
    n1[d] = s_a[x]; n2[d] = s_b[x]; ov[d] = s_ov[x];
    o1[d] = (s_o[x] & 1) ? '-' : '+'; o2[d] = (s_o[x] & 2) ? '-' : '+';
}
__global__ void fill64_kernel(u64* p, u64 n, u64 v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v;
}

inline unsigned grid_for(u64 n) { return (unsigned)((n + 255) / 256); }

}  // namespace

struct EdgeBuffers {
    Buf lh, lv, sh, sv, tmp, cnt, amax, off, s_a, s_b, s_ov, s_o, keep, pos, set, scal, n1, o1, n2, o2, ov;
};
EdgeBuffers* edge_buffers_create() { return new EdgeBuffers(); }
void edge_buffers_destroy(EdgeBuffers* b) { delete b; }

#define EHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

hipError_t build_edges(EdgeBuffers* B, const EdgeNodes& nd, float presimp, hipStream_t s, EdgeResult* out) {
    memset(out, 0, sizeof *out);
    const u64 n = nd.n, n2x = 2 * n;
    if (n == 0) return hipSuccess;
    // ---- listings, sorted by hash (stable: equal hashes stay in listing order)
    EHIP(B->lh.ensure(n2x * 8)); EHIP(B->sh.ensure(n2x * 8)); EHIP(B->lv.ensure(n2x * 4)); EHIP(B->sv.ensure(n2x * 4));
    hipLaunchKernelGGL(list_kernel, dim3(grid_for(n2x)), dim3(256), 0, s, nd, B->lh.as<u64>(), B->lv.as<u32>());
    size_t tb = 0;
    EHIP(rocprim::radix_sort_pairs(nullptr, tb, B->lh.as<u64>(), B->sh.as<u64>(), B->lv.as<u32>(), B->sv.as<u32>(), (size_t)n2x, 0, 64, s));
    size_t tb2 = 0, tb3 = 0;
    EHIP(B->cnt.ensure(n2x * 4)); EHIP(B->amax.ensure(n2x * 2)); EHIP(B->off.ensure(n2x * 8)); EHIP(B->scal.ensure(64));
    EHIP(rocprim::exclusive_scan(nullptr, tb2, B->cnt.as<u32>(), B->off.as<u64>(), (u64)0, (size_t)n2x, rocprim::plus<u64>(), s));
    EHIP(B->tmp.ensure(std::max(tb, tb2) + 256));
    EHIP(rocprim::radix_sort_pairs(B->tmp.p, tb, B->lh.as<u64>(), B->sh.as<u64>(), B->lv.as<u32>(), B->sv.as<u32>(), (size_t)n2x, 0, 64, s));
    // ---- pass 1: candidates per (node, key)
    QueryArgs q; memset(&q, 0, sizeof q);
    q.nd = nd; q.presimp = presimp; q.sh = B->sh.as<u64>(); q.sv = B->sv.as<u32>(); q.n_list = n2x;
    q.cnt = B->cnt.as<u32>(); q.amax = B->amax.as<u16>(); q.off = B->off.as<u64>();
    hipLaunchKernelGGL(query_kernel<false>, dim3(grid_for(n2x)), dim3(256), 0, s, q);
    EHIP(rocprim::exclusive_scan(B->tmp.p, tb2, B->cnt.as<u32>(), B->off.as<u64>(), (u64)0, (size_t)n2x, rocprim::plus<u64>(), s));
    u64 last_off = 0; u32 last_cnt = 0;
    EHIP(hipMemcpyAsync(&last_off, B->off.as<u64>() + (n2x - 1), 8, hipMemcpyDeviceToHost, s));
    EHIP(hipMemcpyAsync(&last_cnt, B->cnt.as<u32>() + (n2x - 1), 4, hipMemcpyDeviceToHost, s));
    EHIP(hipStreamSynchronize(s));
    const u64 n_slots = last_off + last_cnt;
    if (n_slots == 0) return hipSuccess;
    // ---- pass 2: candidate edges at their offsets, presimp removals into a hash set
    u64 set_cap = 1024; while (set_cap < 2 * n_slots + 16) set_cap <<= 1;
    EHIP(B->s_a.ensure(n_slots * 4)); EHIP(B->s_b.ensure(n_slots * 4)); EHIP(B->s_ov.ensure(n_slots * 4)); EHIP(B->s_o.ensure(n_slots));
    EHIP(B->keep.ensure(n_slots * 4)); EHIP(B->pos.ensure(n_slots * 8));
    const bool ps = presimp > 0.0f;
    if (ps) { EHIP(B->set.ensure(set_cap * 8)); hipLaunchKernelGGL(fill64_kernel, dim3(1024), dim3(256), 0, s, B->set.as<u64>(), set_cap, EMPTY64); }
    EHIP(hipMemsetAsync(B->scal.p, 0, 64, s));
    q.s_a = B->s_a.as<u32>(); q.s_b = B->s_b.as<u32>(); q.s_ov = B->s_ov.as<u32>(); q.s_o = B->s_o.as<u8>();
    q.set = B->set.as<u64>(); q.set_mask = set_cap - 1; q.n_removed = (unsigned long long*)B->scal.p;
    hipLaunchKernelGGL(query_kernel<true>, dim3(grid_for(n2x)), dim3(256), 0, s, q);
    // ---- filter (either direction removed), compact in order
    hipLaunchKernelGGL(filter_kernel, dim3(grid_for(n_slots)), dim3(256), 0, s, n_slots, q.s_a, q.s_b, q.s_o, q.set, q.set_mask, ps, B->keep.as<u32>());
    EHIP(rocprim::exclusive_scan(nullptr, tb3, B->keep.as<u32>(), B->pos.as<u64>(), (u64)0, (size_t)n_slots, rocprim::plus<u64>(), s));
    EHIP(B->tmp.ensure(tb3 + 256));
    EHIP(rocprim::exclusive_scan(B->tmp.p, tb3, B->keep.as<u32>(), B->pos.as<u64>(), (u64)0, (size_t)n_slots, rocprim::plus<u64>(), s));
    u64 last_pos = 0, removed = 0; u32 last_keep = 0;
    EHIP(hipMemcpyAsync(&last_pos, B->pos.as<u64>() + (n_slots - 1), 8, hipMemcpyDeviceToHost, s));
    EHIP(hipMemcpyAsync(&last_keep, B->keep.as<u32>() + (n_slots - 1), 4, hipMemcpyDeviceToHost, s));
    EHIP(hipMemcpyAsync(&removed, B->scal.p, 8, hipMemcpyDeviceToHost, s));
    EHIP(hipStreamSynchronize(s));
    const u64 n_edges = last_pos + last_keep;
    EHIP(B->n1.ensure(n_edges * 4 + 4)); EHIP(B->n2.ensure(n_edges * 4 + 4)); EHIP(B->ov.ensure(n_edges * 4 + 4));
    EHIP(B->o1.ensure(n_edges + 4)); EHIP(B->o2.ensure(n_edges + 4));
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n_slots)), dim3(256), 0, s, n_slots, B->keep.as<u32>(), B->pos.as<u64>(), q.s_a, q.s_b, q.s_ov, q.s_o,
                       B->n1.as<u32>(), B->o1.as<u8>(), B->n2.as<u32>(), B->o2.as<u8>(), B->ov.as<u32>());
    EHIP(hipStreamSynchronize(s));
    out->n = n_edges; out->n1 = B->n1.as<u32>(); out->o1 = B->o1.as<u8>(); out->n2 = B->n2.as<u32>(); out->o2 = B->o2.as<u8>();
    out->overlap = B->ov.as<u32>(); out->presimp_removed = removed;
    return hipSuccess;
}

hipError_t sort_segments_u64(EdgeBuffers* B, const uint64_t* keys_in, uint64_t* keys_out, uint64_t n, uint32_t n_segments, const uint32_t* offsets, hipStream_t s) {
    if (!n || !n_segments) return hipSuccess;
    size_t tb = 0;
    EHIP(rocprim::segmented_radix_sort_keys(nullptr, tb, keys_in, keys_out, (unsigned int)n, n_segments, offsets, offsets + 1, 0, 64, s));
    EHIP(B->tmp.ensure(tb + 256));
    EHIP(rocprim::segmented_radix_sort_keys(B->tmp.p, tb, keys_in, keys_out, (unsigned int)n, n_segments, offsets, offsets + 1, 0, 64, s));
    return hipSuccess;
}
