// vt_emu.cpp — host run of rust_mdbg_amd/csrc/vt_core.h (the "vertical" bit-slice prototype of round 5: measured, not integrated; profiles/r05_notes.md section 2).
// TEST INFRASTRUCTURE ONLY (tests/test_emu_cpu.py): 32 machines of one lane walk 32 consecutive segments of ONE sequence (machine i = segment i, warm-up from the two
// segments in front, exactly the prototype kernels' arrangement); the flags they raise are compared with a plain evaluation of the top VT_B bits of both strands'
// ntHash over the homopolymer-compressed sequence: a flag at a kept position p says "the l-mer that ends VT_B - 1 kept positions in front of p has its top VT_B bits
// clear in the forward or the reverse hash" — nothing more and nothing less.
#include <cstdint>
#include <cstring>
#include <vector>

#include "vt_core.h"

typedef uint32_t u32; typedef uint64_t u64;

template <int L>
static long run(const uint8_t* codes /* 2-bit codes, A=0 C=1 T=2 G=3 */, u32 seg_len, u32 n_seg /* <= 32 */, uint8_t* flag_out /* [n_seg * seg_len] */) {
    // vertical registers: V[c][r] bit i = code plane of position r of segment i
    const u32 n = seg_len * n_seg;
    std::vector<u32> v0(seg_len, 0), v1(seg_len, 0);
    for (u32 i = 0; i < n_seg; ++i) for (u32 r = 0; r < seg_len; ++r) { const u32 c = codes[i * seg_len + r]; v0[r] |= (c & 1u) << i; v1[r] |= ((c >> 1) & 1u) << i; }
    VtState<L> S; vt_reset(S);
    u32 pc0 = 0, pc1 = 0;
    // warm-up: the segment in front (bit i <- bit i - 1), delay line only for the first part, chains for the last 24 steps — then the machine's own segment
    for (u32 r = 0; r < seg_len; ++r) {
        const u32 c0 = v0[r] << 1, c1 = v1[r] << 1;
        if (r + 24 < seg_len) vt_step<L, 0>(S, c0, c1, pc0, pc1, 0u); else vt_step<L, 1>(S, c0, c1, pc0, pc1, 0u);
        pc0 = c0; pc1 = c1;
    }
    for (u32 r = 0; r < seg_len; ++r) {
        const u32 cd = vt_step<L, 2>(S, v0[r], v1[r], pc0, pc1, 0u);
        pc0 = v0[r]; pc1 = v1[r];
        for (u32 i = 0; i < n_seg; ++i) flag_out[i * seg_len + r] = (uint8_t)((cd >> i) & 1u);
    }
    return (long)n;
}

extern "C" {
long vt_emu_flags(const uint8_t* codes, uint32_t seg_len, uint32_t n_seg, uint32_t l, uint8_t* flag_out) {
    if (n_seg > 32 || seg_len < 32) return -1;
    switch (l) {
        case 8: return run<8>(codes, seg_len, n_seg, flag_out);
        case 12: return run<12>(codes, seg_len, n_seg, flag_out);
        case 14: return run<14>(codes, seg_len, n_seg, flag_out);
        case 20: return run<20>(codes, seg_len, n_seg, flag_out);
        default: return -2;
    }
}
void vt_emu_transpose32(uint32_t* x) { vt_transpose32(x); }
uint32_t vt_emu_bits(void) { return (uint32_t)VT_B; }
}
