// vt_core.h — "vertical" bit-sliced sketch filter (gfx950), host-compilable like bs_core.h (tests/emu/ runs it on the CPU).
//
// Same arithmetic as bs_core.h (top VT_B bits of the forward / reverse ntHash of every l-mer of the homopolymer-compressed
// read, src/read.rs:157-211 + nthash crate), other data layout.  bs_core.h keeps 32 CONSECUTIVE positions in a word, so every
// "look back u positions" is a funnel shift (41 half-rate v_alignbit per word and strand pair) and the homopolymer compression
// has to squeeze bits inside words first (117 instructions per raw word).  Here a 32-bit register holds ONE position of 32
// independent MACHINES: bit i of every register belongs to machine i, which walks its own contiguous segment of the RAW stream,
// one position per step.  "Look back u kept positions" is then a register name, and homopolymer compression is a CONDITIONAL
// SHIFT of the machine's delay line (one v_bitop3 select per register under the keep plane): no compaction, no dense stream,
// no scan, no funnel shift — every instruction of the step is a full-rate three-input logic op.
//
// One step, per register of 32 machines (c0/c1: code planes of this raw position, pc0/pc1: of the one before):
//   keep   k = (c0 ^ pc0) | (c1 ^ pc1) [| read start]                                (src/read.rs:163-167: run starts)
//   shift  s[d] <- k ? s[d-1] : s[d], d = DMAX..1 (s[0] = this position's code, s[1] <- pc: the code of the last kept position)
//   hash   T-plane of hash bit j sits at ONE depth (forward: d = 63 - j, reverse: d = j - JLO), so out-bit planes are sliding XORs over
//          L consecutive depths: Y(0) = XOR_{d<L} f_d(s[d]), Y(t+1) = Y(t) ^ f_t(s[t]) ^ f_{t+L}(s[t+L]); the plane function is folded
//          into the XOR (acc ^ f(p0, p1) is ONE v_bitop3)
//   chain  Y(t) describes the l-mer that ENDS t kept positions back (forward: hash bit 63 - t; reverse: hash bit 64 - VT_B + t), so the
//          bits of one l-mer meet in an OR chain that advances on kept steps only: C[t] <- k ? (C[t-1] | Y(t)) : C[t]
//   test   cand = k & ~((C_f[VT_B-2] | Y_f(VT_B-1)) & (C_r[VT_B-2] | Y_r(VT_B-1)))  — raised at the kept position VT_B - 1 kept
//          positions behind the END of an l-mer whose forward or reverse hash has its top VT_B bits clear (a NECESSARY condition
//          for hash <= bound when the bound's top VT_B bits are clear; the survivors are re-evaluated exactly, 64 bits).
#pragma once
#include "../../rust_mdbg_amd/csrc/bs_core.h"

constexpr int VT_B = 8;               // hash bits evaluated by the vertical filter

// v_bitop3_b32: result bit = IMM[(a << 2) | (b << 1) | c]
template <unsigned IMM> BS_HD bs_u32 vt_bitop3(bs_u32 a, bs_u32 b, bs_u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, IMM);
#else
    bs_u32 r = 0;
    for (unsigned idx = 0; idx < 8; ++idx) if ((IMM >> idx) & 1u)
        r |= ((idx & 4) ? a : ~a) & ((idx & 2) ? b : ~b) & ((idx & 1) ? c : ~c);
    return r;
#endif
}
// table of acc ^ f(p0, p1), f given by its truth table over the code p0 | p1 << 1 (operands: acc, p0, p1)
constexpr unsigned vt_imm_xorfn(unsigned truth) {
    unsigned imm = 0;
    for (unsigned idx = 0; idx < 8; ++idx) {
        const unsigned a = (idx >> 2) & 1u, p0 = (idx >> 1) & 1u, p1 = idx & 1u;
        imm |= (a ^ ((truth >> (p0 | (p1 << 1))) & 1u)) << idx;
    }
    return imm;
}
template <unsigned TRUTH> BS_HD bs_u32 vt_xorfn_t(bs_u32 acc, bs_u32 p0, bs_u32 p1) { return vt_bitop3<vt_imm_xorfn(TRUTH)>(acc, p0, p1); }
// acc ^ f(p0, p1); truth is a compile-time constant at every call after unrolling (0 and 15 never reach this function)
BS_HD bs_u32 vt_xorfn(bs_u32 acc, bs_u32 p0, bs_u32 p1, unsigned truth) {
    switch (truth & 15u) {
        case 1: return vt_xorfn_t<1>(acc, p0, p1);   case 2: return vt_xorfn_t<2>(acc, p0, p1);   case 3: return vt_xorfn_t<3>(acc, p0, p1);
        case 4: return vt_xorfn_t<4>(acc, p0, p1);   case 5: return vt_xorfn_t<5>(acc, p0, p1);   case 6: return vt_xorfn_t<6>(acc, p0, p1);
        case 7: return vt_xorfn_t<7>(acc, p0, p1);   case 8: return vt_xorfn_t<8>(acc, p0, p1);   case 9: return vt_xorfn_t<9>(acc, p0, p1);
        case 10: return vt_xorfn_t<10>(acc, p0, p1); case 11: return vt_xorfn_t<11>(acc, p0, p1); case 12: return vt_xorfn_t<12>(acc, p0, p1);
        case 13: return vt_xorfn_t<13>(acc, p0, p1); case 14: return vt_xorfn_t<14>(acc, p0, p1);
        default: return acc;
    }
}
// k ? a : b
BS_HD bs_u32 vt_sel(bs_u32 k, bs_u32 a, bs_u32 b) { return vt_bitop3<0xCA>(k, a, b); }

template <int L> struct VtGeo {
    static constexpr int DMAX = L + VT_B - 2;              // deepest delay-line entry
    static constexpr int JLO = 64 - VT_B - L + 1;          // lowest hash bit that reaches an evaluated out-bit
};
template <int L> struct VtState {
    bs_u32 s0[L + VT_B - 1], s1[L + VT_B - 1];             // code planes of the d-th previous KEPT position, d = 1 .. DMAX ([0] unused)
    bs_u32 cf[VT_B - 1], cr[VT_B - 1];                     // OR chains of the two strands
};
template <int L> BS_HD void vt_reset(VtState<L>& S) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int d = 0; d < L + VT_B - 1; ++d) { S.s0[d] = 0; S.s1[d] = 0; }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int t = 0; t < VT_B - 1; ++t) { S.cf[t] = 0; S.cr[t] = 0; }
}

template <int L, bool FWD> BS_HD constexpr unsigned vt_truth(int d) { return FWD ? bs_truth_f(63 - d) : bs_truth_r(VtGeo<L>::JLO + d); }

// the VT_B out-bit planes of one strand from the (already shifted) delay line; y[t] is the plane, bit t of inv says it is complemented
template <int L, bool FWD> BS_HD void vt_strand(const VtState<L>& S, bs_u32 c0, bs_u32 c1, bs_u32 y[VT_B], unsigned& inv) {
    bs_u32 acc = 0; unsigned par = 0; inv = 0;
    auto term = [&](int d) {
        const unsigned tr = vt_truth<L, FWD>(d);
        if (tr == 0) return;
        if (tr == 15) { par ^= 1u; return; }
        acc = vt_xorfn(acc, d ? S.s0[d] : c0, d ? S.s1[d] : c1, tr);
    };
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int d = 0; d < L; ++d) term(d);
    y[0] = acc; inv |= par;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int t = 0; t < VT_B - 1; ++t) { term(t); term(t + L); y[t + 1] = acc; inv |= par << (t + 1); }
}

// MODE 0: delay line only (warm-up); 1: delay line + chains (end of the warm-up); 2: + candidates.
// rs: forced keeps (read starts, src/read.rs: every read is compressed on its own).  Returns the candidate plane (MODE 2).
template <int L, int MODE> BS_HD bs_u32 vt_step(VtState<L>& S, bs_u32 c0, bs_u32 c1, bs_u32 pc0, bs_u32 pc1, bs_u32 rs) {
    constexpr int DMAX = VtGeo<L>::DMAX;
    const bs_u32 k = vt_bitop3<0xF6>(c0 ^ pc0, c1, pc1) | rs;          // (c0 ^ pc0) | (c1 ^ pc1): IMM = a | (b ^ c)
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int d = DMAX; d >= 2; --d) { S.s0[d] = vt_sel(k, S.s0[d - 1], S.s0[d]); S.s1[d] = vt_sel(k, S.s1[d - 1], S.s1[d]); }
    S.s0[1] = vt_sel(k, pc0, S.s0[1]); S.s1[1] = vt_sel(k, pc1, S.s1[1]);
    if (MODE == 0) return 0;
    bs_u32 yf[VT_B], yr[VT_B]; unsigned invf, invr;
    vt_strand<L, true>(S, c0, c1, yf, invf);
    vt_strand<L, false>(S, c0, c1, yr, invr);
    bs_u32 cand = 0;
    if (MODE == 2) {
        // any_f = cf[last] | yf', any_r = cr[last] | yr' (the chains as they stood BEFORE this step); cand = k & ~(any_f & any_r)
        const bs_u32 af = ((invf >> (VT_B - 1)) & 1u) ? vt_bitop3<0xF3>(S.cf[VT_B - 2], yf[VT_B - 1], 0u) : (S.cf[VT_B - 2] | yf[VT_B - 1]);      // a | ~b
        const bs_u32 both = ((invr >> (VT_B - 1)) & 1u) ? vt_bitop3<0xA2>(S.cr[VT_B - 2], yr[VT_B - 1], af) : vt_bitop3<0xA8>(S.cr[VT_B - 2], yr[VT_B - 1], af);   // (a | b') & c
        cand = k & ~both;
    }
    // chains, last stage first (each reads the stage before it as it stood before this step)
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int t = VT_B - 2; t >= 1; --t) {
        const bs_u32 uf = ((invf >> t) & 1u) ? vt_bitop3<0xA2>(S.cf[t - 1], yf[t], k) : vt_bitop3<0xA8>(S.cf[t - 1], yf[t], k);      // (a | b') & k
        S.cf[t] = vt_bitop3<0xF2>(uf, k, S.cf[t]);                                                                                    // a | (~b & c)
        const bs_u32 ur = ((invr >> t) & 1u) ? vt_bitop3<0xA2>(S.cr[t - 1], yr[t], k) : vt_bitop3<0xA8>(S.cr[t - 1], yr[t], k);
        S.cr[t] = vt_bitop3<0xF2>(ur, k, S.cr[t]);
    }
    S.cf[0] = (invf & 1u) ? vt_bitop3<0x3A>(k, yf[0], S.cf[0]) : vt_sel(k, yf[0], S.cf[0]);      // k ? ~y : c
    S.cr[0] = (invr & 1u) ? vt_bitop3<0x3A>(k, yr[0], S.cr[0]) : vt_sel(k, yr[0], S.cr[0]);
    return cand;
}

// ---- 32 x 32 bit transpose: x'[r] bit i = x[i] bit r (rows: the machines' words, columns: positions) --------------------
BS_HD void vt_transpose32(bs_u32 x[32]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 16- and 8-bit stages: byte shuffles (one v_perm per word); 4-, 2-, 1-bit stages: shift + select
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bs_u32 a = x[j], b = x[j + 16];
        x[j] = __builtin_amdgcn_perm(b, a, 0x05040100u); x[j + 16] = __builtin_amdgcn_perm(b, a, 0x07060302u);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) if (!(j & 8)) {
        const bs_u32 a = x[j], b = x[j + 8];
        x[j] = __builtin_amdgcn_perm(b, a, 0x06020400u); x[j + 8] = __builtin_amdgcn_perm(b, a, 0x07030501u);
    }
    BS_VCONST(m4, 0x0F0F0F0F); BS_VCONST(m2, 0x33333333); BS_VCONST(m1, 0x55555555);
#pragma unroll
    for (int j = 0; j < 32; ++j) if (!(j & 4)) { const bs_u32 a = x[j], b = x[j + 4]; x[j] = vt_sel(m4, a, b << 4); x[j + 4] = vt_sel(m4, a >> 4, b); }
#pragma unroll
    for (int j = 0; j < 32; ++j) if (!(j & 2)) { const bs_u32 a = x[j], b = x[j + 2]; x[j] = vt_sel(m2, a, b << 2); x[j + 2] = vt_sel(m2, a >> 2, b); }
#pragma unroll
    for (int j = 0; j < 32; ++j) if (!(j & 1)) { const bs_u32 a = x[j], b = x[j + 1]; x[j] = vt_sel(m1, a, bs_shl1(b)); x[j + 1] = vt_sel(m1, a >> 1, b); }
#else
    static const bs_u32 M[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
    int q = 0;
    for (int s = 16; s >= 1; s >>= 1, ++q)
        for (int j = 0; j < 32; ++j) if (!(j & s)) {
            const bs_u32 a = x[j], b = x[j + s], m = M[q];
            x[j] = (a & m) | ((b << s) & ~m); x[j + s] = ((a >> s) & m) | (b & ~m);
        }
#endif
}
