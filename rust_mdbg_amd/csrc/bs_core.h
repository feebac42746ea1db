// bs_core.h — bit-sliced building blocks of the sketch kernel (gfx950), written so that the same source also
// compiles as plain host C++: tests/emu/ runs these functions on the CPU, lane by lane, against the oracle
// (test infrastructure only; the product never executes them on the host).
//
// Representation.  A raw or homopolymer-compressed ("dense") base stream is held as TWO BIT PLANES, 32 positions
// per 32-bit word, MSB FIRST: position 32*w + i lives in bit (31 - i) of word w.  Plane 0 = bit 0 of the 2-bit
// code, plane 1 = bit 1; code = (ascii >> 1) & 3, i.e. A=0 C=1 T=2 G=3 (complement = code ^ 2).
// MSB-first makes every "look back by u positions" a right funnel shift (one v_alignbit_b32) and lets the
// compaction below count with right shifts, which issue at twice the rate of left shifts on gfx950
// (profiles/r01_g_valu_rates.txt).
//
// What is computed (reference: nthash crate 0.5.1 as called from rust-mdbg src/read.rs:196):
//   fh(e) = XOR_{u<l} rol(h[c(e-u)], u)        forward hash of the l-mer ENDING at dense position e
//   rh(e) = XOR_{u<l} rol(rc[c(e-u)], l-1-u)   reverse-complement hash of the same l-mer
//   selected iff min(fh, rh) <= bound (src/read.rs:196, inclusive).
// The kernel evaluates only the top BS_B bits of fh and rh, bit-sliced over 32 positions per lane, as a
// NECESSARY condition (prefix(hash) <= prefix(bound)); the rare survivors are re-evaluated exactly (64 bits).
//
// Bit b of fh(e) = XOR_u H_{b-u}[e-u] with H_j[q] = bit j of h[c(q)]: a fixed boolean function of the two code
// bits per j.  With T_j = H_j delayed by (63 - j), every out-bit is a sliding XOR over l consecutive T planes:
//   W_b = XOR_{j=b-l+1..b} T_j   and   out_b, delayed by (63 - b), equals W_b.
// So one funnel shift per hash-bit plane j (l + BS_B - 1 of them per strand), one XOR per plane and out-bit, and
// one final per-out-bit alignment (which needs the neighbouring word: a DPP wave shift on the device).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BS_HD __host__ __device__ __forceinline__
#else
#define BS_HD inline
#endif

typedef uint32_t bs_u32;
typedef uint64_t bs_u64;

// ntHash seeds by code (A C T G) and their complements' seeds (T G A C)
#define BS_SEED_A 0x3c8bfbb395c60474ull
#define BS_SEED_C 0x3193c18562a02b4cull
#define BS_SEED_G 0x20323ed082572324ull
#define BS_SEED_T 0x295549f54be24456ull

// hash bits evaluated by the bit-sliced filter (top bits 63 .. 64-BS_B).  A tunable, not a switch: the filter is a NECESSARY condition (prefix(hash) <= prefix(bound)) and
// every survivor is evaluated exactly, so any width gives the same minimizers; fewer planes = fewer filter instructions and more exact evaluations.  8 is the measured
// optimum at the densities of BASELINE.json (profiles/r06_bs_b_sweep.txt: 6 .. 10 at d = 0.002 and 0.003); -DMDBG_BS_B=n builds another width.
#ifndef MDBG_BS_B
#define MDBG_BS_B 8
#endif
constexpr int BS_B = MDBG_BS_B;
static_assert(BS_B >= 3 && BS_B <= 12, "filter width");
constexpr int BS_MAX_L = 32;

BS_HD constexpr bs_u64 bs_seed_f(int code) { return code == 0 ? BS_SEED_A : code == 1 ? BS_SEED_C : code == 2 ? BS_SEED_T : BS_SEED_G; }
BS_HD constexpr bs_u64 bs_seed_r(int code) { return bs_seed_f(code ^ 2); }
// truth table over the code (bit c = value for code c) of hash bit j
BS_HD constexpr bs_u32 bs_truth_f(int j) {
    return (bs_u32)(((bs_seed_f(0) >> j) & 1) | (((bs_seed_f(1) >> j) & 1) << 1) | (((bs_seed_f(2) >> j) & 1) << 2) | (((bs_seed_f(3) >> j) & 1) << 3));
}
BS_HD constexpr bs_u32 bs_truth_r(int j) {
    return (bs_u32)(((bs_seed_r(0) >> j) & 1) | (((bs_seed_r(1) >> j) & 1) << 1) | (((bs_seed_r(2) >> j) & 1) << 2) | (((bs_seed_r(3) >> j) & 1) << 3));
}

BS_HD bs_u32 bs_popc(bs_u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (bs_u32)__popc(x);
#else
    return (bs_u32)__builtin_popcount(x);
#endif
}
// ({hi, lo} >> s) & 0xffffffff, s in 0..31
BS_HD bs_u32 bs_alignbit(bs_u32 hi, bs_u32 lo, bs_u32 s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (bs_u32)(((((bs_u64)hi) << 32) | lo) >> (s & 31));
#endif
}
BS_HD bs_u32 bs_brev(bs_u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
#endif
}
BS_HD bs_u32 bs_xor3(bs_u32 a, bs_u32 b, bs_u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
// a ^ b ^ c where b and/or c are known to be zero planes (the flags are compile-time constants after unrolling): the three-input
// logic builtin is opaque to the optimiser, an XOR with a literal zero would otherwise stay an instruction
BS_HD bs_u32 bs_xor3z(bs_u32 a, bs_u32 b, bool bz, bs_u32 c, bool cz) {
    if (bz && cz) return a;
    if (bz) return a ^ c;
    if (cz) return a ^ b;
    return bs_xor3(a, b, c);
}
// (a ^ ia) | (b ^ ib) | (c ^ ic) with ia/ib/ic = all-ones or zero chosen by the low three bits of inv3 (bit 0: a): one v_bitop3
// (full rate; v_or3_b32 issues at half rate on gfx950, profiles/r01_g_valu_rates.txt) with the complements folded into the table
BS_HD bs_u32 bs_or3i(bs_u32 a, bs_u32 b, bs_u32 c, bs_u32 inv3) {
#if defined(__HIP_DEVICE_COMPILE__)
    switch (inv3 & 7u) {
        case 0: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE);
        case 1: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEF);
        case 2: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFB);
        case 3: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xBF);
        case 4: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFD);
        case 5: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xDF);
        case 6: return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF7);
        default: return __builtin_amdgcn_bitop3_b32(a, b, c, 0x7F);
    }
#else
    return (a ^ ((inv3 & 1u) ? ~0u : 0u)) | (b ^ ((inv3 & 2u) ? ~0u : 0u)) | (c ^ ((inv3 & 4u) ? ~0u : 0u));
#endif
}
// x << 1 as an addition: v_add_u32 issues at full rate, v_lshlrev_b32 at half rate (same table)
BS_HD bs_u32 bs_shl1(bs_u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    bs_u32 r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x << 1;
#endif
}
// per-half left shift of the two 16-bit halves of x by the amounts in the halves of s (each < 16 ... or 16 -> 0 handled by the caller)
BS_HD bs_u32 bs_pk_shl16(bs_u32 x, bs_u32 s) {
#if defined(__HIP_DEVICE_COMPILE__)
    bs_u32 r;
    asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "v"(s), "v"(x));
    return r;
#else
    const bs_u32 lo = ((x & 0xFFFFu) << (s & 15u)) & 0xFFFFu, hi = (((x >> 16) << ((s >> 16) & 15u)) & 0xFFFFu) << 16;
    return hi | lo;
#endif
}
BS_HD constexpr bs_u64 bs_rol64(bs_u64 x, unsigned r) { return (x << (r & 63)) | (x >> ((64 - (r & 63)) & 63)); }

// plane "in[p - u]" of a stream held as (prev2, prev, cur) words, u in 0..63
BS_HD bs_u32 bs_delay(bs_u32 cur, bs_u32 prev, bs_u32 prev2, int u) {
    if (u == 0) return cur;
    if (u < 32) return bs_alignbit(prev, cur, (bs_u32)u);
    if (u == 32) return prev;
    return bs_alignbit(prev2, prev, (bs_u32)(u - 32));
}

// boolean function of the two code planes given by a truth table with bit 0 clear (value for code 0 is 0);
// after inlining with a constant table each case is one v_bitop3 / v_and / v_xor
BS_HD bs_u32 bs_plane0(bs_u32 truth, bs_u32 p0, bs_u32 p1) {
    bs_u32 r = 0;
    if (truth & 2) r |= p0 & ~p1;
    if (truth & 4) r |= ~p0 & p1;
    if (truth & 8) r |= p0 & p1;
    return r;
}

// ---- homopolymer compaction: both planes of one raw word squeezed towards the MSB under keep mask m --------
// Two levels.  (1) Every BYTE is squeezed towards its own MSB by three rounds of Hacker's Delight 7-4 "compress", mirrored (zeros
// are counted from the MSB side, so the parallel prefix uses right shifts, which issue at full rate on gfx950, and only the moves use
// left shifts) and run on the four bytes at once: the prefix XOR stops at byte boundaries (masked shifts), so a bit moves by the number
// of dropped bits in front of it IN ITS BYTE — three rounds instead of five, each with a three-step instead of a five-step prefix.
// (2) The four left-justified bytes are closed up: bytes 1 and 3 move up against bytes 0 and 2 by one packed 16-bit shift (per-half
// amounts 8 - popcount of the byte in front), then the low half moves up against the high half by 16 - popcount of the high half.
// 198 instead of 267 issue cycles per word and plane pair (full-rate ops 2.4, half-rate 4.2 cycles; round 2's five-round network).
// A 32-bit constant held in a VECTOR register: the three-input logic op issues at half rate when one operand is a scalar register
// (profiles/r01_g_valu_rates.txt), which is where the compiler puts a constant it cannot encode inline.
#if defined(__HIP_DEVICE_COMPILE__)
#define BS_VCONST(name, val) bs_u32 name; asm("v_mov_b32 %0, %1" : "=v"(name) : "i"(val))
#else
#define BS_VCONST(name, val) const bs_u32 name = (val)
#endif
BS_HD void bs_compress2(bs_u32 m, bs_u32& x0, bs_u32& x1) {
    BS_VCONST(k7f, 0x7F7F7F7F); BS_VCONST(k3f, 0x3F3F3F3F); BS_VCONST(k0f, 0x0F0F0F0F);
    BS_VCONST(khi8, 0xFF00FF00); BS_VCONST(klo8, 0x00FF00FF); BS_VCONST(khi16, 0xFFFF0000); BS_VCONST(klo16, 0x0000FFFF);
    x0 &= m; x1 &= m;
    const bs_u32 m_in = m;
    bs_u32 mk = (~m >> 1) & k7f;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 3; ++i) {
        bs_u32 mp = mk ^ ((mk >> 1) & k7f);
        mp ^= (mp >> 2) & k3f;
        mp ^= (mp >> 4) & k0f;
        const bs_u32 mv = mp & m;
        const int s = 1 << i;
        const bs_u32 t0 = x0 & mv, t1 = x1 & mv;
        // a move by ONE position is an addition: (x ^ t) | (t << 1) = x - t + 2 t = x + t — the bit in front of a moving bit is either moving itself or
        // empty, so no carry leaves the moving group, and none crosses a byte (a byte's top bit never moves) — one full-rate instruction instead of three
        if (i == 0) { m += mv; x0 += t0; x1 += t1; }
        else { m = (m ^ mv) | (mv << s); x0 = (x0 ^ t0) | (t0 << s); x1 = (x1 ^ t1) | (t1 << s); }
        mk &= ~mp;
    }
    // per-half amounts: high half 8 - popc(byte 0), low half 8 - popc(byte 2); then 16 - popc(bytes 0..1)
    const bs_u32 c0 = bs_popc(m_in & 0xFF000000u), c2 = bs_popc(m_in & 0x0000FF00u), c01 = bs_popc(m_in >> 16);
    const bs_u32 sh8 = ((8u - c0) << 16) | (8u - c2);
    const bs_u32 sh16 = 16u - c01;
    x0 = (x0 & khi8) | bs_pk_shl16(x0 & klo8, sh8);
    x1 = (x1 & khi8) | bs_pk_shl16(x1 & klo8, sh8);
    x0 = (x0 & khi16) | ((x0 & klo16) << sh16);
    x1 = (x1 & khi16) | ((x1 & klo16) << sh16);
}

// position (0 = MSB) of the (n+1)-th set bit of m counted from the MSB; n < popcount(m).
// Binary search on "how many bits are set among the top pos + w positions" — one shift, one popcount, one compare per level, m and n
// are never modified (round 2 renormalised m and n at every level: 9-10 instructions per level instead of 5-6).
BS_HD bs_u32 bs_select_msb(bs_u32 m, bs_u32 n) {
    bs_u32 pos = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int w = 16; w >= 1; w >>= 1) {
        const bs_u32 c = bs_popc(m >> ((bs_u32)(32 - w) - pos));      // set bits among the top pos + w positions (pos + w <= 31 ... 32 - w - pos >= 1 except the first level)
        pos |= n >= c ? (bs_u32)w : 0u;
    }
    return pos;
}

// hash bit j is the same for all four bases: its plane is a constant (folded into `inv`)
template <bool FWD> BS_HD constexpr bool bs_tzero(int j) {
    const bs_u32 t = FWD ? bs_truth_f(j) : bs_truth_r(j);
    return t == 0 || t == 15;
}

// ---- bit-sliced filter: per-strand W planes of one dense word -------------------------------------------------
// Input: planes of the word (c*), of the word before (p*) and two before (q*).  Output W[0..BS_B): W[i] belongs to
// hash bit b = 63 - i; complement constants are folded into `inv` (bit i set: plane i holds the complement).
template <int L, bool FWD>
BS_HD void bs_strand_planes(bs_u32 c0, bs_u32 c1, bs_u32 p0, bs_u32 p1, bs_u32 q0, bs_u32 q1, bs_u32 W[BS_B], bs_u32& inv) {
    constexpr int JLO = 64 - BS_B - L + 1;          // lowest hash bit that reaches an evaluated out-bit
    constexpr int NT = L + BS_B - 1;                // planes T_JLO .. T_63
    bs_u32 T[NT];                                   // normalised planes (truth table with bit 0 clear); complements -> `inv`
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int t = 0; t < NT; ++t) {
        const int j = JLO + t;
        bs_u32 truth = FWD ? bs_truth_f(j) : bs_truth_r(j);
        // forward: base at distance u feeds out-bit b through hash bit j = b - u, T_j is H_j delayed by 63 - j;
        // reverse: j = b - (l-1) + u, T_j is R_j delayed by j - JLO
        const int d = FWD ? 63 - j : j - JLO;
        if (truth & 1) truth ^= 15;
        if (truth == 0) { T[t] = 0; }
        else {
            // the delayed plane is a function of the delayed code planes; build it from the word pair it straddles
            const bs_u32 a_cur = d < 32 ? bs_plane0(truth, c0, c1) : bs_plane0(truth, p0, p1);
            const bs_u32 a_prev = d < 32 ? bs_plane0(truth, p0, p1) : bs_plane0(truth, q0, q1);
            const int dd = d < 32 ? d : d - 32;
            T[t] = dd == 0 ? a_cur : bs_alignbit(a_prev, a_cur, (bs_u32)dd);
        }
    }
    // sliding XOR over L consecutive planes: W_b = XOR_{j=b-L+1..b} T_j (same index set for both strands, other delays);
    // W_63 in full (three-input XORs over the planes that are not constant), then W_{b-1} = W_b ^ T_b ^ T_{b-L}
    inv = 0;
    {
        constexpr int hi0 = NT - 1, lo0 = NT - L;
        bs_u32 w = 0, pend = 0;
        bool have_w = false, have_p = false;            // compile-time constants once the loop is unrolled
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int t = lo0; t <= hi0; ++t) {
            if (bs_tzero<FWD>(JLO + t)) continue;
            if (!have_w) { w = T[t]; have_w = true; }
            else if (!have_p) { pend = T[t]; have_p = true; }
            else { w = bs_xor3(w, pend, T[t]); have_p = false; }
        }
        if (have_p) w ^= pend;
        W[0] = w;
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 1; i < BS_B; ++i) W[i] = bs_xor3z(W[i - 1], T[NT - i], bs_tzero<FWD>(JLO + NT - i), T[NT - i - L], bs_tzero<FWD>(JLO + NT - i - L));
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < BS_B; ++i) {
        const int b = 63 - i;
        bs_u32 par = 0;
        for (int j = b - L + 1; j <= b; ++j) par ^= (FWD ? bs_truth_f(j) : bs_truth_r(j)) & 1;
        inv |= par << i;
    }
}

// final alignment and the bit-sliced comparison "top BS_B bits of the hash <= top BS_B bits of the bound".
// W: this word's planes, Wp: the planes of the word before (neighbouring lane).  bmask[i] is all-ones iff bit
// (63 - i) of the bound is set.  The result is the candidate plane for END positions shifted by BS_B - 1:
// bit for stream position x reports the l-mer ending at x - (BS_B - 1).
// ZERO: the top BS_B bits of the bound are all zero (density < 2^-BS_B, the usual setting): the comparison is
// "all BS_B hash bits are zero", a plain OR-reduction.
template <bool FWD, bool ZERO>
BS_HD bs_u32 bs_strand_compare(const bs_u32 W[BS_B], const bs_u32 Wp[BS_B], bs_u32 inv, const bs_u32 bmask[BS_B]) {
    bs_u32 x[BS_B], xr[BS_B];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < BS_B; ++i) {
        // forward: out_b is W_b delayed by (63 - b) = i already; bring every bit to the common delay BS_B - 1
        const int d = FWD ? (BS_B - 1) - i : i;
        xr[i] = d == 0 ? W[i] : bs_alignbit(Wp[i], W[i], (bs_u32)d);        // complements (inv) are folded into the logic below
    }
    if (ZERO) {
        // OR tree over the BS_B planes, three inputs per v_bitop3 (the first triple, then the running value with two more planes, a last single plane when one is left)
        bs_u32 any = bs_or3i(xr[0], xr[1], xr[2], inv);
        int i = 3;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (; i + 1 < BS_B; i += 2) any = bs_or3i(any, xr[i], xr[i + 1], (inv >> (i - 1)) & 6u);
        if (i < BS_B) any |= ((inv >> i) & 1u) ? ~xr[i] : xr[i];
        return ~any;
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < BS_B; ++i) x[i] = ((inv >> i) & 1) ? ~xr[i] : xr[i];
    bs_u32 le = 0xFFFFFFFFu;
    // from the least significant evaluated bit up: le = bound_bit ? (le | ~x) : (le & ~x)
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = BS_B - 1; i >= 0; --i) {
        const bs_u32 nx = ~x[i];
        le = (le & nx) | (bmask[i] & (le | nx));
    }
    return le;
}

// ---- exact 64-bit evaluation of one candidate -----------------------------------------------------------------
// v0 / v1: the code planes of the 32 dense positions ending at the candidate, bit u = position e - u.
// tab: (1 << 2*GS) x {F, R}: GS-base groups indexed by (GS bits of v0) | (GS bits of v1) << GS, bit v = distance v:
//   F = XOR_v rol(h[c_v], v), R = XOR_v rol(rc[c_v], GS - 1 - v).
// A last group of T = l mod GS bases is looked up like a full one with its missing (farther) positions read as code 0 ('A'); what
// 'A' contributes at those positions is a compile-time constant that is XORed out again (round 2 evaluated the tail base by base
// with select chains over the 64-bit seeds: 14 more registers for l = 14).
template <int GS, int T> BS_HD constexpr bs_u64 bs_tail_f() { bs_u64 k = 0; for (int v = T; v < GS; ++v) k ^= bs_rol64(bs_seed_f(0), (unsigned)v); return k; }
template <int GS, int T> BS_HD constexpr bs_u64 bs_tail_r() { bs_u64 k = 0; for (int v = T; v < GS; ++v) k ^= bs_rol64(bs_seed_r(0), (unsigned)(GS - 1 - v)); return k; }
template <int GS, int L>
BS_HD bs_u64 bs_exact_hash(bs_u32 v0, bs_u32 v1, const bs_u64* tab) {
    bs_u64 fh = 0, rh = 0;
    constexpr int G = L / GS, T = L % GS;
    constexpr bs_u32 M = (1u << GS) - 1u;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < G; ++g) {
        const bs_u32 idx = ((v0 >> (GS * g)) & M) | (((v1 >> (GS * g)) & M) << GS);
        fh ^= bs_rol64(tab[2 * idx], (unsigned)(GS * g));
        rh ^= bs_rol64(tab[2 * idx + 1], (unsigned)(L - GS - GS * g));
    }
    if (T) {
        constexpr bs_u32 MT = (1u << T) - 1u;
        const bs_u32 idx = ((v0 >> (GS * G)) & MT) | (((v1 >> (GS * G)) & MT) << GS);
        fh ^= bs_rol64(tab[2 * idx] ^ bs_tail_f<GS, T>(), (unsigned)(GS * G));
        rh ^= bs_rol64(tab[2 * idx + 1] ^ bs_tail_r<GS, T>(), (unsigned)((64 + T - GS) & 63));       // l - 1 - (GS*G + v) = (T - GS) + (GS - 1 - v)
    }
    return fh < rh ? fh : rh;
}

template <int GS>
inline void bs_make_table(bs_u64* tab /* 2 << 2*GS */) {
    for (int idx = 0; idx < (1 << (2 * GS)); ++idx) {
        bs_u64 f = 0, r = 0;
        for (int v = 0; v < GS; ++v) {
            const int c = (((idx >> (GS + v)) & 1) << 1) | ((idx >> v) & 1);
            f ^= bs_rol64(bs_seed_f(c), (unsigned)v);
            r ^= bs_rol64(bs_seed_r(c), (unsigned)(GS - 1 - v));
        }
        tab[2 * idx] = f; tab[2 * idx + 1] = r;
    }
}
constexpr int BS_GS = 3;             // group size of the exact tables: 64 entries x 16 B = 1 KB of LDS.  (4: three lookups instead of four for l = 12, but 4 KB — the
                                     // tile kernel lost 10 %: 26.9 KB per workgroup no longer run six to a CU, profiles/r04_l_micro_ab.txt)
