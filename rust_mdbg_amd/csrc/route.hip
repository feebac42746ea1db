// route.hip — key-range routing of k-min-mer occurrences and cross-rank resolution.
//
// Reads shard by record: each rank sketches its own reads.  Every k-min-mer occurrence becomes a record
// (canonical key[k], global ordinal) owned by rank mulhi64(keyhash, world); records are bucketed by owner here,
// exchanged with ONE all-to-all (RCCL, driven by rust_mdbg_amd/dist.py) and inserted into the owner's table.
// DbgEntry.index (order of first sighting over ALL keys, src/main.rs:598,661) and the A-th sighting's
// seqlen/shift (src/main.rs:680-684) depend on other ranks' data: the owner sends the ordinal to the rank that
// generated that read (`resolve_*` below run there) and gets ranks / metadata back.
#include "mdbg_dev.h"

struct RouteArgs {
    const u64* mh; const u32* mread; const u64* roff; u64 i0, i1; u32 slot0; u64 first_ordinal; u32 k; u32 world;
    u64* counts;      // [world] running counts (COUNT pass) / write cursors (WRITE pass)
    u64* out;         // records, (k+1) u64 each
};

template <bool WRITE>
__global__ __launch_bounds__(256) void route_kernel(RouteArgs a) {
    __shared__ u32 lcnt[64];
    __shared__ u64 lbase[64];
    if (threadIdx.x < 64) lcnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 i = a.i0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = a.k;
    bool valid = false; u32 dest = 0, my = 0; bool rev = false; u64 ord = 0;
    const u64* w = a.mh + i;
    if (i < a.i1) {
        const u32 slot = a.mread[i];
        const u64 rs = a.roff[slot], re = a.roff[slot + 1];
        if (re - rs > k && i + k <= re && i - rs <= WIN_MASK) {
            valid = true;
            rev = window_reversed(w, k);
            const u64 h = key_hash_window(w, k, rev);
            dest = (u32)__umul64hi(h, (u64)a.world);
            ord = ((a.first_ordinal + (slot - a.slot0)) << WIN_BITS) | (i - rs);
            my = atomicAdd(&lcnt[dest], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.world && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd((unsigned long long*)&a.counts[threadIdx.x], (unsigned long long)lcnt[threadIdx.x]);
    if (!WRITE) return;
    __syncthreads();
    if (valid) {
        u64* o = a.out + (lbase[dest] + my) * (k + 1);
        for (u32 j = 0; j < k; ++j) o[j] = rev ? w[k - 1 - j] : w[j];
        o[k] = ord;
    }
}

// compact list of occupied slots: m1, A-th ordinal (~0 when the node fails the abundance filter), count, slot id.
// One allocation atomic per 1024-thread block (same-address atomics serialise).
__global__ __launch_bounds__(1024) void export_kernel(FinArgs F, u64* __restrict__ counter,
                                                      u64* __restrict__ o_m1, u64* __restrict__ o_ma, u32* __restrict__ o_count, u64* __restrict__ o_slot) {
    const Slot* tab = F.tab; const u64 cap = F.cap;
    __shared__ u32 wcnt[16];
    __shared__ u64 bbase;
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    bool occ = false; Slot e{};
    if (s < cap) { e = tab[s]; occ = e.word != EMPTY; }
    const u64 m = __ballot(occ);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wcnt[wv] = (u32)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0;
        for (int i = 0; i < 16; ++i) { const u32 c = wcnt[i]; wcnt[i] = tot; tot += c; }
        bbase = tot ? atomicAdd((unsigned long long*)counter, (unsigned long long)tot) : 0;
    }
    __syncthreads();
    if (occ) {
        const u64 idx = bbase + wcnt[wv] + __popcll(m & ((1ull << lane) - 1));
        const SlotView v = slot_view(e, s, F.mx, F.A, rep_ordinal(F, e.word));
        o_m1[idx] = v.first; o_ma[idx] = v.solid ? v.ath : EMPTY; o_count[idx] = v.count; o_slot[idx] = s;
    }
}

// generator side: mark queried first-sighting ordinals in the dense bitmaps ...
__global__ void resolve_mark_kernel(FinArgs F, const u64* __restrict__ ord, const u8* __restrict__ solid, u64 n) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    u64 i, D; decode_ordinal(F, ord[q], i, D);
    atomicOr((unsigned long long*)&F.bm_first[D >> 6], 1ull << (D & 63));
    if (solid[q]) atomicOr((unsigned long long*)&F.bm_solid[D >> 6], 1ull << (D & 63));
}
// ... and answer with the local ranks
__global__ void resolve_rank_kernel(FinArgs F, const u64* __restrict__ ord, u64 n, u64* __restrict__ rank_first, u64* __restrict__ rank_solid) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    u64 i, D; decode_ordinal(F, ord[q], i, D);
    const u64 below = (1ull << (D & 63)) - 1;
    rank_first[q] = F.pre_first[D >> 6] + __popcll(F.bm_first[D >> 6] & below);
    rank_solid[q] = F.pre_solid[D >> 6] + __popcll(F.bm_solid[D >> 6] & below);
}
// metadata of the sighting with ordinal ord[q]: 6 u64 = {seqlen | reversed << 32, shift0, shift1, src_read, src_start, src_end}
__global__ void resolve_meta_kernel(FinArgs F, const u64* __restrict__ ord, u64 n, u64* __restrict__ meta) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u64 oa = ord[q];
    u64* o = meta + q * 6;
    if (oa == EMPTY) { for (int j = 0; j < 6; ++j) o[j] = 0; return; }
    const u32 k = F.k;
    u64 i, D; decode_ordinal(F, oa, i, D);
    const u64* w = F.mh + i; const u32* p = F.mpos + i;
    const bool rev = window_reversed(w, k);
    const u64 first = p[1] - p[0], last = p[k - 1] - p[k - 2];
    o[0] = (u64)(u32)((u64)p[k - 1] + 1 - p[0] + 1) | ((u64)(rev ? 1 : 0) << 32);
    o[1] = rev ? last : first; o[2] = rev ? first : last;
    o[3] = oa >> WIN_BITS; o[4] = p[0]; o[5] = (u64)p[k - 1] + F.l;
}
// canonical keys of the given slots
__global__ void slot_keys_kernel(const Slot* __restrict__ tab, KeySrc ks, const u64* __restrict__ slots, u64 n, u64* __restrict__ keys) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u64 w = tab[slots[q]].word;
    for (u32 j = 0; j < ks.k; ++j) keys[q * ks.k + j] = rep_elem(ks, w, j);
}

void launch_route(const RouteArgs& a, bool write, hipStream_t s) {
    if (a.i1 <= a.i0) return;
    const unsigned nb = (unsigned)((a.i1 - a.i0 + 255) / 256);
    if (write) hipLaunchKernelGGL(route_kernel<true>, dim3(nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(route_kernel<false>, dim3(nb), dim3(256), 0, s, a);
}
void launch_export(const FinArgs& F, u64* counter, u64* o_m1, u64* o_ma, u32* o_count, u64* o_slot, hipStream_t s) {
    hipLaunchKernelGGL(export_kernel, dim3((unsigned)((F.cap + 1023) / 1024)), dim3(1024), 0, s, F, counter, o_m1, o_ma, o_count, o_slot);
}
void launch_resolve_mark(const FinArgs& F, const u64* ord, const u8* solid, u64 n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(resolve_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, ord, solid, n);
}
void launch_resolve_rank(const FinArgs& F, const u64* ord, u64 n, u64* rf, u64* rs, hipStream_t s) {
    if (n) hipLaunchKernelGGL(resolve_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, ord, n, rf, rs);
}
void launch_resolve_meta(const FinArgs& F, const u64* ord, u64 n, u64* meta, hipStream_t s) {
    if (n) hipLaunchKernelGGL(resolve_meta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, ord, n, meta);
}
void launch_slot_keys(const Slot* tab, const KeySrc& ks, const u64* slots, u64 n, u64* keys, hipStream_t s) {
    if (n) hipLaunchKernelGGL(slot_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tab, ks, slots, n, keys);
}
