// route.hip — key-range routing of k-min-mer occurrences and cross-rank resolution.
//
// Reads shard by record: each rank sketches its own reads.  Every k-min-mer occurrence becomes a record
// (canonical key[k], global ordinal, key hash) owned by rank mulhi64(keyhash, world); records are bucketed by owner here,
// exchanged with ONE all-to-all (RCCL, driven by rust_mdbg_amd/dist.py) and inserted into the owner's table.
// DbgEntry.index (order of first sighting over ALL keys, src/main.rs:598,661) and the A-th sighting's
// seqlen/shift (src/main.rs:680-684) depend on other ranks' data: the owner sends the ordinal to the rank that
// generated that read (`resolve_*` below run there) and gets ranks / metadata back.
#include "mdbg_dev.h"

// One routed record = k + 2 u64: canonical key[k], global ordinal, key hash (the owner probes with it, no re-hashing).
struct RouteArgs {
    const u64* mh; const u32* mread; const u64* roff; u64 i0, i1; u32 slot0; u64 first_ordinal; u32 k; u32 world;
    u64* hbuf; u8* wflag;        // per minimizer index: key hash, flags (bit0 valid window, bit1 reversed)  [pass 1 -> pass 2]
    u32* blk_cnt;                // [n_blocks][world] records of block b for destination d (pass 1), no atomics
    const u64* blk_off;          // [n_blocks][world] first record index of block b inside bucket d (absolute, pass 2)
    u32 blk0;                    // index of this launch's first block in blk_cnt / blk_off
    u64* out;                    // bucketed records
};
constexpr int ROUTE_THREADS = 1024;

// pass 1: orientation + key hash + destination of every window; per-block, per-destination counts
__global__ __launch_bounds__(ROUTE_THREADS) void route_count_kernel(RouteArgs a) {
    __shared__ u32 lcnt[64];
    if (threadIdx.x < 64) lcnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 i = a.i0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = a.k;
    if (i < a.i1) {
        const u32 slot = a.mread[i];
        const u64 rs = a.roff[slot], re = a.roff[slot + 1];
        u8 fl = 0;
        if (re - rs > k && i + k <= re && i - rs <= WIN_MASK) {
            const u64* w = a.mh + i;
            const bool rev = window_reversed(w, k);
            const u64 h = key_hash_window(w, k, rev);
            a.hbuf[i] = h;
            fl = 1 | (rev ? 2 : 0);
            atomicAdd(&lcnt[(u32)__umul64hi(h, (u64)a.world)], 1u);
        }
        a.wflag[i] = fl;
    }
    __syncthreads();
    if (threadIdx.x < a.world) a.blk_cnt[(size_t)(a.blk0 + blockIdx.x) * a.world + threadIdx.x] = lcnt[threadIdx.x];
}

// blk_off[b][d] = bucket_base[d] + sum_{b' < b} blk_cnt[b'][d]; one workgroup per destination; totals[d] = bucket size
__global__ __launch_bounds__(1024) void route_scan_kernel(const u32* __restrict__ blk_cnt, u32 n_blocks, u32 world, u64* __restrict__ blk_off, u64* __restrict__ totals) {
    __shared__ u64 ws[16]; __shared__ u64 run;
    const u32 d = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < n_blocks; b0 += 1024) {
        const u32 b = b0 + tid;
        const u64 v = b < n_blocks ? blk_cnt[(size_t)b * world + d] : 0;
        u64 inc = v;
        for (int s = 1; s < 64; s <<= 1) { const u64 t = __shfl_up(inc, s, 64); if (lane >= s) inc += t; }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        u64 base = run, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) base += ws[q]; tot += ws[q]; }
        if (b < n_blocks) blk_off[(size_t)b * world + d] = base + inc - v;      // relative to the bucket start (made absolute on the host side of pass 2)
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) totals[d] = run;
}

// pass 2: every window gets its slot in its bucket; each 8(k+2)-byte row is written by consecutive lanes (coalesced)
__global__ __launch_bounds__(ROUTE_THREADS) void route_write_kernel(RouteArgs a, const u64* __restrict__ bucket_base) {
    __shared__ u32 lcnt[64];
    if (threadIdx.x < 64) lcnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 i = a.i0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = a.k, RS = k + 2;
    u8 fl = 0; u64 h = 0, ord = 0, pos = 0;
    if (i < a.i1) fl = a.wflag[i];
    if (fl & 1) {
        h = a.hbuf[i];
        const u32 dest = (u32)__umul64hi(h, (u64)a.world);
        const u32 slot = a.mread[i];
        ord = ((a.first_ordinal + (slot - a.slot0)) << WIN_BITS) | (i - a.roff[slot]);
        pos = bucket_base[dest] + a.blk_off[(size_t)(a.blk0 + blockIdx.x) * a.world + dest] + atomicAdd(&lcnt[dest], 1u);
    }
    const int lane = threadIdx.x & 63;
    u64 vmask = __ballot(fl & 1);
    while (vmask) {
        const int src = __ffsll((unsigned long long)vmask) - 1;
        vmask &= vmask - 1;
        const u64 i_s = __shfl(i, src, 64), pos_s = __shfl(pos, src, 64), ord_s = __shfl(ord, src, 64), h_s = __shfl(h, src, 64);
        const bool rev_s = (__shfl((int)fl, src, 64) & 2) != 0;
        u64* row = a.out + pos_s * RS;
        for (u32 j = lane; j < RS; j += 64) {
            u64 v;
            if (j < k) v = a.mh[i_s + (rev_s ? k - 1 - j : j)];
            else v = j == k ? ord_s : h_s;
            row[j] = v;
        }
    }
}

// ---- owner-side export, grouped by the rank that has to answer ---------------------------------------
// Every distinct k-min-mer asks the rank that generated its FIRST sighting for its index / row (query 1, list A), every
// solid one also asks the rank of its A-th sighting for seqlen / shift / origin (query 2, list S).  Both lists come out
// bucketed by answering rank (spans of read ordinals per rank are passed in), so the driver can exchange them as they are.
struct ExportArgs {
    FinArgs F;
    const u64* span_lo; const u32* span_rank; u32 n_spans; u32 world;    // read-ordinal spans sorted by start
    u32* blk_cnt;                // [n_blocks][2*world]: counts for list A then list S
    const u64* blk_off;          // [n_blocks][2*world]
    const u64* bucket_base;      // [2*world]
    u64* a_first; u8* a_solid;                                           // list A
    u64* s_ath; u64* s_slot; u32* s_count; u64* s_idx1;                  // list S (s_idx1 = position of the entry in list A)
};
__device__ inline u32 span_owner(const ExportArgs& a, u64 ord) {
    const u64 ro = ord >> WIN_BITS;
    u32 lo = 0, hi = a.n_spans - 1;
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if (a.span_lo[mid] <= ro) lo = mid; else hi = mid - 1; }
    return a.span_rank[lo];
}
template <bool WRITE>
__global__ __launch_bounds__(1024) void export_grouped_kernel(ExportArgs a) {
    __shared__ u32 lcnt[128];
    if (threadIdx.x < 128) lcnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    bool occ = false; SlotView v{}; u32 g1 = 0, g2 = 0, r1 = 0, r2 = 0;
    if (s < a.F.cap) {
        const Slot e = a.F.tab[s];
        if (e.word != EMPTY) {
            occ = true;
            v = slot_view(e, s, a.F.mx, a.F.casc, a.F.A, rep_ordinal(a.F, e.word), a.F.ath_override);
            g1 = span_owner(a, v.first);
            r1 = atomicAdd(&lcnt[g1], 1u);
            if (v.solid) { g2 = span_owner(a, v.ath); r2 = atomicAdd(&lcnt[a.world + g2], 1u); }
        }
    }
    __syncthreads();
    const u32 W2 = 2 * a.world;
    if (!WRITE) { if (threadIdx.x < W2) a.blk_cnt[(size_t)blockIdx.x * W2 + threadIdx.x] = lcnt[threadIdx.x]; return; }
    if (occ) {
        const u64 pa = a.bucket_base[g1] + a.blk_off[(size_t)blockIdx.x * W2 + g1] + r1;
        a.a_first[pa] = v.first; a.a_solid[pa] = v.solid ? 1 : 0;
        if (v.solid) {
            const u64 ps = a.bucket_base[a.world + g2] + a.blk_off[(size_t)blockIdx.x * W2 + a.world + g2] + r2;
            a.s_ath[ps] = v.ath; a.s_slot[ps] = s; a.s_count[ps] = v.count; a.s_idx1[ps] = pa;
        }
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

u32 route_blocks(u64 i0, u64 i1) { return i1 > i0 ? (u32)((i1 - i0 + ROUTE_THREADS - 1) / ROUTE_THREADS) : 0; }
void launch_route_count(const RouteArgs& a, hipStream_t s) {
    if (a.i1 > a.i0) hipLaunchKernelGGL(route_count_kernel, dim3(route_blocks(a.i0, a.i1)), dim3(ROUTE_THREADS), 0, s, a);
}
void launch_route_scan(const u32* blk_cnt, u32 n_blocks, u32 world, u64* blk_off, u64* totals, hipStream_t s) {
    hipLaunchKernelGGL(route_scan_kernel, dim3(world), dim3(1024), 0, s, blk_cnt, n_blocks, world, blk_off, totals);
}
void launch_route_write(const RouteArgs& a, const u64* bucket_base, hipStream_t s) {
    if (a.i1 > a.i0) hipLaunchKernelGGL(route_write_kernel, dim3(route_blocks(a.i0, a.i1)), dim3(ROUTE_THREADS), 0, s, a, bucket_base);
}
void launch_export_grouped(const ExportArgs& a, bool write, hipStream_t s) {
    const unsigned nb = (unsigned)((a.F.cap + 1023) / 1024);
    if (write) hipLaunchKernelGGL(export_grouped_kernel<true>, dim3(nb), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(export_grouped_kernel<false>, dim3(nb), dim3(1024), 0, s, a);
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
