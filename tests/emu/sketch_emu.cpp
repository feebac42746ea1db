// sketch_emu.cpp — host emulation of the tile algorithm of rust_mdbg_amd/csrc/sketch.hip.
//
// TEST INFRASTRUCTURE ONLY (tests/test_emu_cpu.py): runs the device kernel's arithmetic (bs_core.h, the very same
// header the kernel includes) on the CPU, thread by thread and phase by phase with the kernel's data layout, so that
// the bit-sliced filter, the compaction, the dense-stream packing, the candidate bookkeeping and the coordinate
// maps can be compared with the oracle without a GPU.  It is not a fallback: nothing in rust_mdbg_amd/ loads it.
// Tiles the kernel would hand to its generic walker are reported in *n_slow and evaluated by a plain
// restatement of that walker.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../rust_mdbg_amd/csrc/bs_core.h"

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

namespace {
constexpr int RW = 1024, HALO_BASES = 256, HW = HALO_BASES / 32, TT = 256, WPT = RW / TT, DPAD = 4;
constexpr int TILE_STRIDE = RW * 32 - HALO_BASES;

u64 rol64(u64 x, unsigned r) { r &= 63; return (x << r) | (x >> ((64 - r) & 63)); }
u64 nt_h(u8 c) { return c == 'A' ? BS_SEED_A : c == 'C' ? BS_SEED_C : c == 'G' ? BS_SEED_G : c == 'T' ? BS_SEED_T : 0; }
u64 nt_rc(u8 c) { return c == 'A' ? BS_SEED_T : c == 'C' ? BS_SEED_G : c == 'G' ? BS_SEED_C : c == 'T' ? BS_SEED_A : 0; }
bool in_hpc_set(u8 c) { return c == 'A' || c == 'C' || c == 'T' || c == 'G' || c == 'a' || c == 'c' || c == 't' || c == 'g' || c == 'N' || c == 'n'; }

u32 find_read(const u64* off, u32 lo, u32 hi, u64 p) {
    while (lo < hi) { u32 mid = lo + ((hi - lo + 1) >> 1); if (off[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}
u32 range_mask(int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > 32) b = 32;
    if (a >= b) return 0;
    const u32 from_a = 0xFFFFFFFFu >> (u32)a;
    const u32 from_b = b == 32 ? 0u : 0xFFFFFFFFu >> (u32)b;
    return from_a & ~from_b;
}

struct Out { std::vector<u64> hash; std::vector<u32> pos, read; };

bool walk_lmer(const u8* b, bool hpc, u64 rlo, u64 p, u32 l, u64& start, u64& hash) {
    u64 q = p, fh = 0, rh = 0;
    for (int j = (int)l - 1;; --j) {
        const u8 c = b[q];
        fh ^= rol64(nt_h(c), l - 1 - j);
        rh ^= rol64(nt_rc(c), j);
        if (j == 0) break;
        if (q == rlo) return false;
        u64 q2 = q - 1;
        if (hpc) { const u8 c2 = b[q2]; if (in_hpc_set(c2)) while (q2 > rlo && b[q2 - 1] == c2) --q2; }
        q = q2;
    }
    start = q; hash = fh < rh ? fh : rh;
    return true;
}

template <int L>
void tile(const u8* bases, int64_t nb, const u64* offsets, u32 n_reads, bool hpc, u64 bound, const u64* t4, u32 gt, Out& out, u64* n_slow, u64* n_cand_total) {
    const int64_t raw0 = (int64_t)gt * TILE_STRIDE - HALO_BASES;
    const int64_t first_base = (int64_t)offsets[0];
    std::vector<u32> dense(2 * (DPAD + RW + 4), 0), kw(RW), force(RW, 0), cand(RW + 8, 0);
    std::vector<u16> rpre(RW + 8, 0), cpre(RW + 8, 0);
    for (u32 r = 0; r < n_reads; ++r) { const int64_t rel = (int64_t)offsets[r] - raw0; if (rel >= 0 && rel < RW * 32) force[rel >> 5] |= 0x80000000u >> (rel & 31); }
    // phase 1: planes (MSB first), alphabet
    std::vector<u32> X0(RW, 0), X1(RW, 0);
    bool bad = false;
    for (int i = 0; i < RW * 32; ++i) {
        const int64_t q = raw0 + i;
        u8 c = 'A';
        if (q >= 0 && q < nb) c = bases[q];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') bad = true;
        const u32 code = (c >> 1) & 3;
        if (code & 1) X0[i >> 5] |= 0x80000000u >> (i & 31);
        if (code & 2) X1[i >> 5] |= 0x80000000u >> (i & 31);
    }
    // phase 2
    const int64_t lo = first_base - raw0, hi = nb - raw0;
    u32 off = 0, Hh = 0;
    for (int w = 0; w < RW; ++w) {
        if (w == HW) Hh = off;
        const u32 vm = range_mask(lo - 32 * (int64_t)w, hi - 32 * (int64_t)w);
        u32 x0 = X0[w], x1 = X1[w], k = vm;
        if (hpc) {
            const u32 p0 = w ? X0[w - 1] : 0, p1 = w ? X1[w - 1] : 0;
            const u32 d0 = bs_alignbit(p0, x0, 1), d1 = bs_alignbit(p1, x1, 1);
            k = ((x0 ^ d0) | (x1 ^ d1) | force[w]) & vm;
            if (w == 0) k |= 0x80000000u & vm;
            bs_compress2(k, x0, x1);
        } else { x0 &= vm; x1 &= vm; if (vm != 0xFFFFFFFFu) bs_compress2(k, x0, x1); }
        kw[w] = k; rpre[w] = (u16)off;
        const u32 n = bs_popc(k);
        if (n) {
            const u32 wi = off >> 5, s = off & 31;
            dense[2 * (DPAD + wi)] |= x0 >> s; dense[2 * (DPAD + wi) + 1] |= x1 >> s;
            if (s + n > 32) { dense[2 * (DPAD + wi + 1)] |= bs_alignbit(x0, 0u, s); dense[2 * (DPAD + wi + 1) + 1] |= bs_alignbit(x1, 0u, s); }
        }
        off += n;
    }
    const u32 H = off;
    rpre[RW] = (u16)H;
    const bool true_start = raw0 <= first_base;
    if (bad || (!true_start && Hh < (u32)L)) {
        ++*n_slow;
        const u64 t_lo = (u64)gt * TILE_STRIDE; u64 t_hi = t_lo + TILE_STRIDE; if ((int64_t)t_hi > nb) t_hi = (u64)nb;
        for (u64 p = t_lo; p < t_hi; ++p) {
            if ((int64_t)p < first_base) continue;
            const u32 r = find_read(offsets, 0, n_reads - 1, p); const u64 rlo = offsets[r];
            const bool kept = !hpc || p == rlo || !(bases[p] == bases[p - 1] && in_hpc_set(bases[p]));
            u64 start, h;
            if (kept && walk_lmer(bases, hpc, rlo, p, L, start, h) && h <= bound) { out.hash.push_back(h); out.pos.push_back((u32)(start - rlo)); out.read.push_back(r); }
        }
        return;
    }
    // phase 3: lanes are words; "neighbour lane" = word D - 1
    u32 bmask[BS_B];
    const u32 btop = (u32)(bound >> (64 - BS_B));
    for (int i = 0; i < BS_B; ++i) bmask[i] = ((btop >> (BS_B - 1 - i)) & 1u) ? 0xFFFFFFFFu : 0u;
    const u32 n_out = H ? ((H + BS_B - 2) >> 5) + 1 : 0;
    {
        u32 Wf_prev[BS_B] = {0}, Wr_prev[BS_B] = {0};
        for (int D = -1; D < (int)n_out; ++D) {
            const u32* dw = dense.data() + 2 * (DPAD + D);
            u32 q0 = 0, q1 = 0;
            if (L + BS_B - 2 >= 32) { q0 = dw[-4]; q1 = dw[-3]; }
            u32 Wf[BS_B], Wr[BS_B], invf, invr;
            bs_strand_planes<L, true>(dw[0], dw[1], dw[-2], dw[-1], q0, q1, Wf, invf);
            bs_strand_planes<L, false>(dw[0], dw[1], dw[-2], dw[-1], q0, q1, Wr, invr);
            if (D >= 0) cand[D] = btop == 0 ? (bs_strand_compare<true, true>(Wf, Wf_prev, invf, bmask) | bs_strand_compare<false, true>(Wr, Wr_prev, invr, bmask))
                                       : (bs_strand_compare<true, false>(Wf, Wf_prev, invf, bmask) | bs_strand_compare<false, false>(Wr, Wr_prev, invr, bmask));
            memcpy(Wf_prev, Wf, sizeof Wf); memcpy(Wr_prev, Wr, sizeof Wr);
        }
    }
    // phase 4
    const u32 e_lo = Hh > (u32)(L - 1) ? Hh : (u32)(L - 1);
    for (int D = 0; D <= RW; ++D)
        cand[D] = (u32)D < n_out ? cand[D] & range_mask((int64_t)e_lo + BS_B - 1 - 32 * (int64_t)D, (int64_t)H + BS_B - 1 - 32 * (int64_t)D) : 0u;
    const float raw_per_dense = (float)RW / (float)(H ? H : 1u);        // the kernel's first guess of the raw word (sketch.hip, dense_to_raw)
    auto dense_to_raw = [&](u32 r) -> u32 {
        u32 w = (u32)((float)r * raw_per_dense);
        w = w < (u32)RW - 1u ? w : (u32)RW - 1u;
        while (rpre[w] > r) --w;
        while (rpre[w + 1] <= r) ++w;
        return 32 * w + bs_select_msb(kw[w], r - rpre[w]);
    };
    for (int D = 0; D <= RW; ++D) {
        u32 w = cand[D];
        while (w) {
            const u32 b = (u32)__builtin_clz(w); w &= ~(0x80000000u >> b);
            ++*n_cand_total;
            const u32 e = 32 * D + b - (BS_B - 1);
            const u32 wi = e >> 5, s = e & 31;
            const u32* dw = dense.data() + 2 * (DPAD + wi);
            const u32 v0 = bs_alignbit(dw[-2], dw[0], 31 - s), v1 = bs_alignbit(dw[-1], dw[1], 31 - s);
            const u64 h = bs_exact_hash<BS_GS, L>(v0, v1, t4);
            if (h > bound) continue;
            // the kernel's rule (sketch.hip, place): the FIRST base decides the read; an l-mer that reaches the next read's start is dropped
            const u32 sd = e - (u32)(L - 1);
            const int64_t abs_start = raw0 + dense_to_raw(sd);
            const u32 r = find_read(offsets, 0, n_reads - 1, (u64)abs_start);
            const int64_t q0 = (int64_t)offsets[r];
            const int64_t q1 = (r + 1 < n_reads ? (int64_t)offsets[r + 1] : nb) - raw0;
            if (r + 1 < n_reads && q1 < (int64_t)RW * 32) {
                const u32 w2 = (u32)q1 >> 5, b2 = (u32)q1 & 31;
                const u32 ds = rpre[w2] + (b2 ? bs_popc(kw[w2] >> (32 - b2)) : 0u);
                if (ds <= e) continue;
            }
            out.hash.push_back(h); out.pos.push_back((u32)(abs_start - q0)); out.read.push_back(r);
        }
    }
}

typedef void (*TileFn)(const u8*, int64_t, const u64*, u32, bool, u64, const u64*, u32, Out&, u64*, u64*);
template <int L> TileFn pick(u32 l) { if (l == (u32)L) return &tile<L>; if constexpr (L > 2) return pick<L - 1>(l); else return nullptr; }
}  // namespace

extern "C" {
// returns the number of minimizers (or -1 on bad l); arrays must hold cap entries
int64_t emu_sketch(const u8* bases, u64 n_bases, const u64* offsets, u64 n_reads, u32 l, double density, int hpc,
                   u64* out_hash, u32* out_pos, u32* out_read, u64 cap, u64* n_slow, u64* n_cand) {
    TileFn fn = pick<32>(l);
    if (!fn || !n_reads) return fn ? 0 : -1;
    const double v = density * 18446744073709551616.0;
    const u64 bound = !(v > 0.0) ? 0 : (v >= 18446744073709551616.0 ? ~0ull : (u64)v);
    u64 t4[2 << (2 * BS_GS)]; bs_make_table<BS_GS>(t4);
    Out out; *n_slow = 0; *n_cand = 0;
    const u64 n_tiles = (n_bases + TILE_STRIDE - 1) / TILE_STRIDE;
    for (u64 t = 0; t < n_tiles; ++t) fn(bases, (int64_t)n_bases, offsets, (u32)n_reads, hpc != 0, bound, t4, (u32)t, out, n_slow, n_cand);
    const u64 n = out.hash.size();
    for (u64 i = 0; i < n && i < cap; ++i) { out_hash[i] = out.hash[i]; out_pos[i] = out.pos[i]; out_read[i] = out.read[i]; }
    return (int64_t)n;
}
// plain unit entry points for the primitives
void emu_compress2(u32 m, u32* x0, u32* x1) { bs_compress2(m, *x0, *x1); }
u32 emu_select_msb(u32 m, u32 n) { return bs_select_msb(m, n); }
}
