// gz_inflate.h — gzip input for the host reader (include/mdbg_emit.h: mdbg_reader_*), without zlib's inflate.
//
// What it replaces: the reference reads ".gz" input through flate2's read::GzDecoder on one thread (src/main.rs:34,173: it decodes the FIRST
// member, checks its CRC and silently ignores whatever follows — so a bgzip'ed or concatenated file loses every read behind its first member there).
// This reader decodes EVERY member (what `zcat` prints; a deliberate superset); zlib's gzread did the same here at ~0.2 GB/s of text per thread, which made the inflate the bound of any run
// that starts from a compressed file.  This decoder (RFC 1951 / 1952, written against the RFCs) keeps a 64-bit bit buffer that is
// refilled without a branch, resolves a literal / length code with ONE lookup in an 11-bit table (longer codes: a second lookup in a
// subtable), copies matches eight bytes at a time, and checks bounds once per iteration instead of once per byte (a careful per-byte
// loop takes over near the ends of the input and of the output window).  The CRC-32 of every member is verified (carry-less multiply
// where the CPU has it).  BGZF files (bgzip: every member is an independent block of at most 64 KiB that carries its compressed size
// in an extra field) are inflated by several threads at once.  An ORDINARY gzip stream has no such entry points and stays on one thread: rounds 3 - 4 carried
// a speculative several-thread decoder for it (pieces entered at block headers found by search, decoded without their history: the pugz / rapidgzip scheme);
// on the GPU box's host it delivered 520 - 550 MB/s of text with 4 - 32 threads against 633 MB/s on ONE (profiles/r04_a_host_reader_gz.txt), it was switched
// off, and round 5 removed it (git history: gz_inflate.h of round 4).  bgzip is the parallel format.
//
// The compressed file is mapped, so the decoder never runs out of input in the middle of a symbol; output is produced in chunks into
// a window that keeps the last 32 KiB as history.  Untrusted input: every table index is masked, every distance is checked against the
// history that exists, every length against the room that is left; a malformed stream ends in `bad`, never in a wild access.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#if defined(__x86_64__)             // carry-less multiplication for the CRC and BMI2 clones of the decoder: x86 hosts only; elsewhere the portable paths
#include <immintrin.h>
#define GZ_X86 1
#else
#define GZ_X86 0
#endif
#include <functional>
#include <zlib.h>                   // crc32_combine only

// function clones are resolved through ifuncs, which run before ThreadSanitizer's runtime is up: none in such builds
#if defined(__SANITIZE_THREAD__) || !GZ_X86
#define GZ_CLONES
#else
#define GZ_CLONES __attribute__((target_clones("bmi2", "default")))
#endif

namespace gz {
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
// vectors whose resize does not fill: the buffers below grow by tens of MB and are written front to back right after
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    template <class U, class... A> void construct(U* p, A&&... a) { if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<u8, NoInit<u8>> Bytes;

// ---- CRC-32 (IEEE 802.3, reflected; RFC 1952 section 8) -----------------------------------------------------------------------
struct CrcTables {
    u32 t[8][256];
    CrcTables() {
        for (u32 i = 0; i < 256; ++i) { u32 c = i; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); t[0][i] = c; }
        for (u32 i = 0; i < 256; ++i) for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
inline const CrcTables& crc_tables() { static const CrcTables T; return T; }
// raw register update (no pre / post inversion): eight bytes per step
inline u32 crc_raw_tables(u32 c, const u8* p, size_t n) {
    const CrcTables& T = crc_tables();
    while (n >= 8) {
        u64 w; memcpy(&w, p, 8);
        w ^= c;
        c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
            T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
    return c;
}
// Folding with carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", 2009;
// constants for the reflected polynomial 0x1DB710641): 64 bytes per step, then 16, then the Barrett reduction.  n >= 64, multiple of 16.
#if GZ_X86
__attribute__((target("pclmul,sse4.1"))) inline u32 crc_raw_clmul(u32 c, const u8* p, size_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll);
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16)), x3 = _mm_loadu_si128((const __m128i*)(p + 32)),
            x4 = _mm_loadu_si128((const __m128i*)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64; n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00),
                      a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        const __m128i b1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), b2 = _mm_clmulepi64_si128(x2, k1k2, 0x11), b3 = _mm_clmulepi64_si128(x3, k1k2, 0x11),
                      b4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(a1, b1), _mm_loadu_si128((const __m128i*)p));
        x2 = _mm_xor_si128(_mm_xor_si128(a2, b2), _mm_loadu_si128((const __m128i*)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(a3, b3), _mm_loadu_si128((const __m128i*)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(a4, b4), _mm_loadu_si128((const __m128i*)(p + 48)));
        p += 64; n -= 64;
    }
#define GZ_FOLD(x, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k3k4, 0x00), _mm_clmulepi64_si128(x, k3k4, 0x11)), next)      /* (a lambda would not inherit the target attribute) */
    x1 = GZ_FOLD(x1, x2); x1 = GZ_FOLD(x1, x3); x1 = GZ_FOLD(x1, x4);
    while (n >= 16) { const __m128i nx = _mm_loadu_si128((const __m128i*)p); x1 = GZ_FOLD(x1, nx); p += 16; n -= 16; }
#undef GZ_FOLD
    // 128 -> 64 bits (this also appends the 32 zero bits of the CRC's definition), 64 -> 32 with R5, then Barrett
    __m128i t = _mm_clmulepi64_si128(k3k4, x1, 0x01);                 // k4 * low half
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    __m128i x2b = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), x2b);
    x2b = x1;
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, poly, 0x10);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, poly, 0x00);
    x1 = _mm_xor_si128(x1, x2b);
    return (u32)_mm_extract_epi32(x1, 1);
}
inline bool have_clmul() { static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return ok; }
#else
inline bool have_clmul() { return false; }
inline u32 crc_raw_clmul(u32 c, const u8*, size_t) { return c; }      // (never called)
#endif
// zlib's convention: crc32(0, ...) of the empty string is 0; chainable
inline u32 crc32(u32 crc, const u8* p, size_t n) {
    u32 c = ~crc;
    if (n >= 128 && have_clmul()) { const size_t m = n & ~(size_t)15; c = crc_raw_clmul(c, p, m); p += m; n -= m; }
    return ~crc_raw_tables(c, p, n);
}

// ---- raw deflate --------------------------------------------------------------------------------------------------------------
// table entry (round 6 layout).  Bits 0-5: ALL the bits the entry consumes — the code AND the extra bits of a length or distance (a subtable pointer: the primary index
// bits) — so that the bit buffer moves on with ONE shift whose amount is the entry's low bits as they are (x86 shifts take the count modulo 64: no mask).  The decoder is bound by
// the chain load -> shift -> load (DNA text deflates to a stream of SHORT MATCHES — any 4- to 8-mer recurs within 32 KiB —: 7.8 M matches of 4.6 bytes in 36 MB of FASTQ at
// level 1, one litlen and one distance lookup each); with the extra bits taken from a copy of the buffer beside the chain a match costs two shifts instead of four.
// Bits 6-7: number of literals of a literal entry (1..3; 0: not a literal entry), their bytes in bits 8-31: where the codes are short one lookup in the primary table delivers
// up to three literals (pack_literals).  Other entries: bits 8-10 kind, 11-14 code bits (BASE: the extra bits start behind them; SUB: index bits of the subtable), 15-31 value
// (base length / base distance / first entry of the subtable).
enum { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
constexpr u32 LITF = 0xC0;                      // (e & LITF) != 0: a literal entry
constexpr u32 entry(u32 kind, u32 nbits, u32 extra, u32 value) {
    return kind == K_LIT ? (nbits | 1u << 6 | value << 8) : kind == K_SUB ? (nbits | kind << 8 | extra << 11 | value << 15) : ((nbits + extra) | kind << 8 | nbits << 11 | value << 15);
}
inline u32 e_tot(u32 e) { return e & 63; }
inline u32 e_nlit(u32 e) { return e >> 6 & 3; }
inline u32 e_kind(u32 e) { return (e & LITF) ? (u32)K_LIT : (e >> 8 & 7); }
inline u32 e_cb(u32 e) { return e >> 11 & 15; }                          // BASE: code bits; SUB: index bits of the subtable
inline u32 e_code_bits(u32 e) { return (e & LITF) || (e >> 8 & 7) != K_BASE ? (e & 63) : (e >> 11 & 15); }      // bits of the CODE (careful loop: extra bits are taken separately)
inline u32 e_value(u32 e) { return e >> 15; }
inline u32 e_extra(u32 e) { return (e >> 8 & 7) == K_SUB ? (e >> 11 & 15) : (e & 63) - (e >> 11 & 15); }      // BASE: extra bits; SUB: index bits
constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr int LIT_TAB = (1 << LIT_BITS) + 288 * 16, DIST_TAB = (1 << DIST_BITS) + 32 * 128;

struct Inflater {
    // input (whole stream in memory)
    const u8* in = nullptr; size_t in_n = 0, ip = 0;
    u64 bb = 0; u32 bc = 0;                     // bit buffer: the low bc bits are unread stream bits (the fast path may hold more above them)
    // block state
    enum St { HEADER, STORED, HUFF, END } st = HEADER;
    bool last = false; u32 stored_left = 0;
    u32 pend_len = 0, pend_dist = 0;            // a match cut by the end of the output chunk
    u32 pend_lit = 0, pend_nlit = 0;            // literals of a packed entry cut by it
    u32 lit[LIT_TAB], dist[DIST_TAB];
    // the first lookup of the fast loop: lit[]'s primary entries, and where a length (its code and its extra bits) and the distance code behind it fit the index together
    // a PAIR entry that decodes both — one load and one shift per match on the decoder's dependent chain instead of two and two (pack_pairs).  32-bit entries: with the
    // 32 KiB of history the decoder's hot set (this table 8 KB, the distance table, the window) stays inside a 48-KB L1
    u32 fast[1 << LIT_BITS];
    const char* err = nullptr;

    void start(const u8* p, size_t n, size_t at) { in = p; in_n = n; ip = at; bb = 0; bc = 0; st = HEADER; last = false; stored_left = 0; pend_len = pend_dist = 0; pend_lit = pend_nlit = 0; err = nullptr; }
    size_t bitpos() const { return ip * 8 - bc; }               // of the next unread bit (bytes the fast path holds above bc are not counted in ip)
    bool fail(const char* m) { err = m; return false; }

    // -- careful bit access (headers, ends of input / output): only counted bits
    void normalize() { if (bc < 64) bb &= (1ull << bc) - 1; }                       // drop what the fast path holds above bc
    bool need(u32 n) { while (bc < n) { if (ip >= in_n) return false; bb |= (u64)in[ip++] << bc; bc += 8; } return true; }      // n <= 32
    u32 take(u32 n) { const u32 v = (u32)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; }
    // first byte of the stream not yet consumed, after dropping the bits up to the next byte boundary (stored blocks, member trailer)
    void byte_align() { normalize(); const u32 drop = bc & 7; bb >>= drop; bc -= drop; ip -= bc >> 3; bb = 0; bc = 0; }

    // canonical Huffman code -> lookup table (codes are read LSB first: entries are indexed by the bit-reversed code)
    // kind_of(sym, &extra, &value) describes a symbol.  Oversubscribed codes are rejected, and so are incomplete ones as zlib does it (inftrees.c): the code of the
    // code lengths must be complete, the other two may be incomplete only as a single code of one bit (or empty); what may stay unused holds K_BAD entries.
    template <class F>
    bool build(const u8* lens, int n, u32* tab, int P, int cap, F kind_of, bool code_lengths = false) {
        u16 count[16] = {0}; u16 next[16];
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        count[0] = 0;
        int left = 1;
        int maxlen = 0;
        for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return fail("oversubscribed Huffman code"); if (count[l]) maxlen = l; }
        if (left > 0 && maxlen != 0 && (code_lengths || maxlen != 1)) return fail("incomplete Huffman code");
        u32 code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = (u16)code; }
        const u32 PM = (1u << P) - 1;
        for (u32 i = 0; i <= PM; ++i) tab[i] = entry(K_BAD, 1, 0, 0);
        // longest code behind every primary prefix (sizes of the subtables)
        u8 sub_bits[1 << LIT_BITS];
        bool any_long = false;
        for (int l = P + 1; l <= 15; ++l) if (count[l]) any_long = true;
        if (any_long) memset(sub_bits, 0, (size_t)PM + 1);
        u16 rev_code[320];
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            u32 c = next[l]++, r = 0;
            for (int b = 0; b < l; ++b) { r = r << 1 | (c & 1); c >>= 1; }
            rev_code[s] = (u16)r;
            if (l > P) { u8& sb = sub_bits[r & PM]; sb = std::max<u8>(sb, (u8)(l - P)); }
        }
        u32 top = PM + 1;
        if (any_long) for (u32 i = 0; i <= PM; ++i) if (sub_bits[i]) {
            const u32 sz = 1u << sub_bits[i];
            if (top + sz > (u32)cap) return fail("Huffman table overflow");
            tab[i] = entry(K_SUB, (u32)P, sub_bits[i], top);
            for (u32 j = 0; j < sz; ++j) tab[top + j] = entry(K_BAD, 1, 0, 0);
            top += sz;
        }
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            u32 extra = 0, value = 0;
            const u32 kind = kind_of(s, extra, value);
            const u32 r = rev_code[s];
            if (l <= P) {
                if (e_kind(tab[r]) == K_SUB) return fail("Huffman code is not prefix free");
                const u32 e = entry(kind, (u32)l, extra, value);
                for (u32 i = r; i <= PM; i += 1u << l) tab[i] = e;
            } else {
                const u32 pe = tab[r & PM];
                if (e_kind(pe) != K_SUB) return fail("Huffman code is not prefix free");
                const u32 sb = e_extra(pe), base = e_value(pe), sl = (u32)(l - P);
                const u32 e = entry(kind, sl, extra, value);
                for (u32 i = r >> P; i < (1u << sb); i += 1u << sl) tab[base + i] = e;
            }
        }
        return true;
    }
    static u32 lit_kind(int s, u32& extra, u32& value) {
        static const u16 base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const u8 ext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) { value = (u32)s; return K_LIT; }
        if (s == 256) return K_EOB;
        if (s > 285) return K_BAD;                                 // 286, 287: in the fixed code, never valid in data
        extra = ext[s - 257]; value = base[s - 257];
        return K_BASE;
    }
    static u32 dist_kind(int s, u32& extra, u32& value) {
        static const u16 base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        if (s > 29) return K_BAD;
        extra = s < 4 ? 0 : (u32)(s - 2) >> 1; value = base[s];
        return K_BASE;
    }

    bool block_header() {
        normalize();
        if (!need(3)) return fail("truncated deflate stream");
        last = take(1) != 0;
        const u32 type = take(2);
        if (type == 0) {
            byte_align();
            if (ip + 4 > in_n) return fail("truncated stored block");
            const u32 len = in[ip] | (u32)in[ip + 1] << 8, nlen = in[ip + 2] | (u32)in[ip + 3] << 8;
            if ((len ^ nlen) != 0xFFFF) return fail("stored block length check failed");
            ip += 4; stored_left = len; st = STORED;
            return true;
        }
        if (type == 3) return fail("reserved block type");
        u8 lens[320];
        int hlit, hdist;
        if (type == 1) {
            hlit = 288; hdist = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            if (!need(14)) return fail("truncated deflate stream");
            hlit = (int)take(5) + 257; hdist = (int)take(5) + 1;
            const int hclen = (int)take(4) + 4;
            if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
            static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            u8 cl[19] = {0};
            for (int i = 0; i < hclen; ++i) { if (!need(3)) return fail("truncated deflate stream"); cl[order[i]] = (u8)take(3); }
            u32 cltab[128];
            if (!build(cl, 19, cltab, 7, 128, [](int s, u32&, u32& value) { value = (u32)s; return (u32)K_LIT; }, true)) return false;
            int i = 0;
            while (i < hlit + hdist) {
                if (!need(7 + 7)) { if (!need(1)) return fail("truncated deflate stream"); }      // (the tail of the stream may hold fewer than 14 bits)
                const u32 e = cltab[bb & 127];
                if (e_kind(e) != K_LIT) return fail("invalid code length code");
                if (e_tot(e) > bc) return fail("truncated deflate stream");
                take(e_tot(e));
                const u32 sym = e >> 8 & 0xFF;
                if (sym < 16) { lens[i++] = (u8)sym; continue; }
                u32 rep, val = 0;
                if (sym == 16) { if (i == 0) return fail("repeat without a previous length"); if (!need(2)) return fail("truncated deflate stream"); val = lens[i - 1]; rep = 3 + take(2); }
                else if (sym == 17) { if (!need(3)) return fail("truncated deflate stream"); rep = 3 + take(3); }
                else { if (!need(7)) return fail("truncated deflate stream"); rep = 11 + take(7); }
                if (i + (int)rep > hlit + hdist) return fail("code length repeat overruns");
                while (rep--) lens[i++] = (u8)val;
            }
            if (lens[256] == 0) return fail("no end-of-block code");
            // the distance lengths follow the literal / length ones directly: move them to their own array position
            memmove(lens + 288, lens + hlit, (size_t)hdist);
        }
        if (!build(lens, hlit, lit, LIT_BITS, LIT_TAB, lit_kind)) return false;
        if (!build(lens + 288, hdist, dist, DIST_BITS, DIST_TAB, dist_kind)) return false;
        pack_literals();
        pack_pairs();
        st = HUFF;
        return true;
    }
    // PAIR entry: bits 0-5 all bits consumed (length code and extra bits + distance code + distance extra bits), 8-10 = K_PAIR, 11-18 the length - 3, 19-23 the distance
    // SYMBOL (base and number of extra bits: dist_base / dist_xbits, two small tables read beside the chain), 24-27 the bits in front of the distance's extra bits.
    // Like pack_literals: the index bits above the length read as the distance code only when that code does not reach beyond the index.
    enum { K_PAIR = 5 };
    static const u16* dist_base_tab() { static const u16 t[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0}; return t; }
    static const u8* dist_xbits_tab() { static const u8 t[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0}; return t; }
    void pack_pairs() {
        constexpr u32 PM = (1u << LIT_BITS) - 1, DM = (1u << DIST_BITS) - 1;
        for (u32 i = 0; i <= PM; ++i) {
            const u32 e = lit[i];
            u32 f = e;
            if (!(e & LITF) && (e >> 8 & 7) == K_BASE && (e & 63) < (u32)LIT_BITS) {             // a length whose code AND extra bits lie inside the index: it is known here
                const u32 l1 = e & 63, lcb = e_cb(e);
                const u32 len = e_value(e) + ((i >> lcb) & ((1u << (l1 - lcb)) - 1));
                const u32 d = dist[(i >> l1) & DM];                                                   // (bits above the index read as zero)
                if (!(d & LITF) && (d >> 8 & 7) == K_BASE && l1 + e_cb(d) <= (u32)LIT_BITS) {
                    // the distance symbol from the entry's base and extra-bit count: symbols 0..3 have bases 1..4, behind them two symbols per extra-bit count x with
                    // bases 1 + (2 << x) and 1 + (3 << x) (RFC 1951 3.2.5)
                    const u32 dx = (d & 63) - e_cb(d), base = e_value(d);
                    const u32 sym = dx ? 2 * dx + 2 + (((base - 1) >> dx) & 1u) : base - 1;
                    f = (l1 + (d & 63)) | (u32)K_PAIR << 8 | (len - 3) << 11 | sym << 19 | (l1 + e_cb(d)) << 24;
                }
            }
            fast[i] = f;
        }
    }

    // Primary entries of short literal codes take the literals that follow along: index bits above the first code that decide a second (and a
    // third) literal completely become part of the entry.  Works on a copy of the single-symbol entries, so a packed entry never feeds another.
    void pack_literals() {
        constexpr u32 PM = (1u << LIT_BITS) - 1;
        static thread_local u32 one[1 << LIT_BITS];
        memcpy(one, lit, sizeof one);
        for (u32 i = 0; i <= PM; ++i) {
            const u32 e1 = one[i];
            if (!(e1 & LITF)) continue;
            const u32 l1 = e_tot(e1);
            if (l1 + 1 > (u32)LIT_BITS) continue;
            const u32 e2 = one[i >> l1];                                  // the unknown bits above read as zero: valid only if the code found does not reach them
            const u32 l2 = e_tot(e2);
            if (!(e2 & LITF) || l1 + l2 > (u32)LIT_BITS) continue;
            u32 bytes = (e1 >> 8 & 0xFF) | (e2 >> 8 & 0xFF) << 8, n = 2, bits = l1 + l2;
            const u32 e3 = one[i >> bits];
            const u32 l3 = e_tot(e3);
            if ((e3 & LITF) && bits + l3 <= (u32)LIT_BITS) { bytes |= (e3 >> 8 & 0xFF) << 16; n = 3; bits += l3; }
            lit[i] = bits | n << 6 | bytes << 8;
        }
    }

    // Decodes until the output chunk [.., out_end) is full, the stream's final block has ended (st == END), or an error (false).
    // `lo` = first byte of valid history in `out`; `op` = write position (in / out).
    GZ_CLONES                                                // (shrx / bzhi for the variable shifts and masks of the loop where the CPU has them: ~7 %)
    bool run(u8* out, size_t lo, size_t& op_io, size_t out_end) {
        size_t op = op_io;
        constexpr u32 LM = (1u << LIT_BITS) - 1, DM = (1u << DIST_BITS) - 1;
        for (;;) {
            while (pend_nlit && op < out_end) { out[op++] = (u8)pend_lit; pend_lit >>= 8; --pend_nlit; }      // literals of a packed entry cut by the previous chunk's end
            if (pend_nlit) { op_io = op; return true; }
            if (pend_len) {                                            // rest of a match cut by the previous chunk's end
                while (pend_len && op < out_end) { out[op] = out[op - pend_dist]; ++op; --pend_len; }
                if (pend_len) { op_io = op; return true; }
            }
            if (st == END) { op_io = op; return true; }
            if (st == HEADER) {
                if (!block_header()) return false;
                continue;
            }
            if (st == STORED) {
                const size_t n = std::min<size_t>(std::min<size_t>(stored_left, out_end - op), in_n - ip);
                memcpy(out + op, in + ip, n); op += n; ip += n; stored_left -= (u32)n;
                if (stored_left) { if (ip >= in_n) return fail("truncated stored block"); op_io = op; return true; }      // chunk full
                st = last ? END : HEADER;
                continue;
            }
            // ---- Huffman block: fast loop while both ends are far away
            bool eob = false;
            if (in_n >= 32 && out_end >= 320) {
                const size_t in_fast = in_n - 32, out_fast = out_end - 320;
                const u16* const dbase = dist_base_tab(); const u8* const dxb = dist_xbits_tab();
                u64 b = bb; u32 c = bc; size_t i = ip;
#define GZ_REFILL() do { u64 w_; memcpy(&w_, in + i, 8); b |= w_ << c; i += (63 - c) >> 3; c |= 56; } while (0)
// (b0: the buffer in front of the entry's bits — the extra bits of a length / distance are read from it beside the chain)
#define GZ_IS_SUB(e) (((e) & (LITF | 7u << 8)) == (u32)K_SUB << 8)
#define GZ_LOOKUP(e) do { e = lit[b & LM]; if (GZ_IS_SUB(e)) { b >>= LIT_BITS; c -= LIT_BITS; e = lit[e_value(e) + (b & ((1u << e_cb(e)) - 1))]; } b0 = b; b >>= (e & 63); c -= e & 63; } while (0)
#define GZ_PUT(e) do { const u32 v_ = e >> 8; memcpy(out + op, &v_, 4); op += e >> 6 & 3; } while (0)
                while (i <= in_fast && op <= out_fast) {
                    GZ_REFILL();
                    u32 e; u64 b0;
                    u32 len; size_t dd;
                    const u32 f = fast[b & LM];
                    if ((f & (LITF | 7u << 8)) == (u32)K_PAIR << 8) {      // a short match in ONE lookup: length and distance code together, the distance's base and extra bits beside the chain
                        b0 = b; b >>= (f & 63); c -= f & 63;
                        len = (f >> 11 & 255) + 3;
                        const u32 sym = f >> 19 & 31;
                        dd = (size_t)dbase[sym] + (size_t)((b0 >> (f >> 24 & 15)) & ((1ull << dxb[sym]) - 1));
                    } else {
                    e = f;
                    if (GZ_IS_SUB(e)) { b >>= LIT_BITS; c -= LIT_BITS; e = lit[e_value(e) + (b & ((1u << e_cb(e)) - 1))]; }
                    b0 = b; b >>= (e & 63); c -= e & 63;
                    if (e & LITF) {                                    // up to three lookups per refill (3 x 15 bits), each up to three literals
                        GZ_PUT(e);
                        GZ_LOOKUP(e);
                        if (e & LITF) {
                            GZ_PUT(e);
                            GZ_LOOKUP(e);
                            if (e & LITF) { GZ_PUT(e); continue; }
                        }
                        GZ_REFILL();                                   // (idempotent: the bits above c are the stream's own)
                    }
                    const u32 kind = e >> 8 & 7;
                    if (kind != K_BASE) { if (kind == K_EOB) { eob = true; break; } bb = b; bc = c; ip = i; return fail("invalid literal / length code"); }
                    const u32 lcb = e_cb(e);                           // the length: base + the (tot - code) bits behind the code, read from the buffer as it was
                    len = e_value(e) + (u32)((b0 >> lcb) & ((1u << ((e & 63) - lcb)) - 1));
                    u32 d = dist[b & DM];                              // (>= 36 bits are left on either way here: a distance takes at most 15 + 13)
                    if (GZ_IS_SUB(d)) { b >>= DIST_BITS; c -= DIST_BITS; d = dist[e_value(d) + (b & ((1u << e_cb(d)) - 1))]; }
                    b0 = b; b >>= (d & 63); c -= d & 63;
                    if ((d & LITF) || (d >> 8 & 7) != K_BASE) { bb = b; bc = c; ip = i; return fail("invalid distance code"); }
                    const u32 dcb = e_cb(d);
                    dd = e_value(d) + (size_t)((b0 >> dcb) & ((1ull << ((d & 63) - dcb)) - 1));
                    }
                    if (dd > op - lo) { bb = b; bc = c; ip = i; return fail("distance reaches in front of the data"); }
                    u8* dst = out + op; const u8* src = dst - dd;
                    op += len;
                    if (dd >= 8) {                                     // sixteen bytes without asking (most matches are shorter), then eight at a time
                        u64 w; memcpy(&w, src, 8); memcpy(dst, &w, 8); memcpy(&w, src + 8, 8); memcpy(dst + 8, &w, 8);
                        for (u32 k = 16; k < len; k += 8) { memcpy(&w, src + k, 8); memcpy(dst + k, &w, 8); }
                    } else if (dd == 1) {
                        memset(dst, *src, len);
                    } else {
                        for (u32 k = 0; k < len; ++k) dst[k] = src[k];
                    }
                }
#undef GZ_REFILL
#undef GZ_LOOKUP
#undef GZ_IS_SUB
#undef GZ_PUT
                bb = b; bc = c; ip = i;
            }
            if (!eob) {
                // ---- careful loop: one table entry at a time, only counted bits, every byte checked
                normalize();
                for (;;) {                                             // (a full chunk is noticed by the symbol that does not fit: an end-of-block code behind the last byte is still taken)
                    need(32);                                          // as many as there are
                    u32 e = lit[bb & LM];
                    u32 used = 0;
                    if (e_kind(e) == K_SUB) { used = LIT_BITS; e = lit[e_value(e) + ((bb >> LIT_BITS) & ((1u << e_cb(e)) - 1))]; }
                    used += e_code_bits(e);
                    const u32 kind = e_kind(e);
                    if (kind == K_BAD || kind == K_SUB) return fail(used > bc ? "truncated deflate stream" : "invalid literal / length code");
                    if (used > bc) return fail("truncated deflate stream");
                    take(used);
                    if (kind == K_LIT) {
                        u32 n = e_nlit(e), v = e >> 8;
                        while (n && op < out_end) { out[op++] = (u8)v; v >>= 8; --n; }
                        if (n) { pend_nlit = n; pend_lit = v; op_io = op; return true; }      // chunk full: the rest comes first thing in the next call
                        continue;
                    }
                    if (kind == K_EOB) { eob = true; break; }
                    const u32 xb = e_extra(e);
                    if (!need(xb)) return fail("truncated deflate stream");
                    const u32 len = e_value(e) + take(xb);
                    need(32);
                    u32 d = dist[bb & DM];
                    used = 0;
                    if (e_kind(d) == K_SUB) { used = DIST_BITS; d = dist[e_value(d) + ((bb >> DIST_BITS) & ((1u << e_cb(d)) - 1))]; }
                    used += e_code_bits(d);
                    if (e_kind(d) != K_BASE) return fail(used > bc ? "truncated deflate stream" : "invalid distance code");
                    if (used > bc) return fail("truncated deflate stream");
                    take(used);
                    const u32 db = e_extra(d);
                    if (!need(db)) return fail("truncated deflate stream");
                    const size_t dd = e_value(d) + take(db);
                    if (dd > op - lo) return fail("distance reaches in front of the data");
                    u32 k = len;
                    while (k && op < out_end) { out[op] = out[op - dd]; ++op; --k; }
                    if (k) { pend_len = k; pend_dist = (u32)dd; op_io = op; return true; }
                }
            }
            st = last ? END : HEADER;
        }
    }

};

// ---- gzip members over a mapped file ----------------------------------------------------------------------------------------------
struct Member { size_t data = 0; bool bgzf = false; u32 bsize = 0; };      // data: first byte of the deflate stream; bsize: whole member, BGZF only
// parses the member header at `at`; false: not a gzip member (or truncated header)
inline bool parse_header(const u8* p, size_t n, size_t at, Member& m) {
    if (at + 10 > n || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8) return false;
    const u8 flg = p[at + 3];
    if (flg & 0xE0) return false;                                      // reserved flag bits
    size_t q = at + 10;
    m.bgzf = false; m.bsize = 0;
    if (flg & 4) {
        if (q + 2 > n) return false;
        const size_t xlen = p[q] | (size_t)p[q + 1] << 8; q += 2;
        if (q + xlen > n) return false;
        for (size_t x = q; x + 4 <= q + xlen;) {
            const size_t sl = p[x + 2] | (size_t)p[x + 3] << 8;
            if (x + 4 + sl > q + xlen) break;
            if (p[x] == 'B' && p[x + 1] == 'C' && sl == 2) { m.bgzf = true; m.bsize = (p[x + 4] | (u32)p[x + 5] << 8) + 1; }
            x += 4 + sl;
        }
        q += xlen;
    }
    if (flg & 8) { while (q < n && p[q]) ++q; if (q >= n) return false; ++q; }
    if (flg & 16) { while (q < n && p[q]) ++q; if (q >= n) return false; ++q; }
    if (flg & 2) {                                                     // CRC-16 of the header (zlib and flate2 check it)
        if (q + 2 > n) return false;
        if ((crc32(0, p + at, q - at) & 0xFFFF) != (p[q] | (u32)p[q + 1] << 8)) return false;
        q += 2;
    }
    m.data = q;
    return true;
}

// Streaming reader of a mapped gzip file: read() like gzread.  With threads > 1 and a BGZF file, groups of blocks are inflated in parallel.
struct GzIn {
    const u8* in = nullptr; size_t n = 0, at = 0;                      // mapped file; at = next member header (between members)
    int threads = 1;
    bool in_member = false, bad = false, done = false, is_bgzf = false;
    const char* err = nullptr;
    Inflater* inf = nullptr;
    Bytes win; size_t lo = 0, rd = 0, wr = 0;                         // window: history from lo, unread bytes [rd, wr)
    u32 crc = 0; u64 produced = 0;                                     // of the member being decoded
    size_t mhist = 0;                                                  // fill(): bytes of the member's history that lie right in front of the write position (<= HIST)
    static constexpr size_t HIST = 32768, CHUNK = 1u << 20;

    ~GzIn() { delete inf; }
    void open(const u8* p, size_t size, int nthreads) {
        in = p; n = size; at = 0; threads = std::max(1, nthreads);
        Member m;
        is_bgzf = parse_header(in, n, 0, m) && m.bgzf;
        inf = new Inflater();
        win.resize(HIST + CHUNK + 64);
    }
    bool fail(const char* m) { bad = true; err = m; return false; }

    // next member at `at`, or the end of the data.  Behind a complete member only zero bytes (block padding of tapes and some archivers) end the data
    // quietly; anything else that is not a member is an error.  This is deliberately STRICTER than the reference: its flate2 read::GzDecoder
    // (src/main.rs:34,173) stops after the first member and never looks at the rest, and zlib's gzread would stop at the damage without a word —
    // either way the reads behind a damaged magic would be lost with MDBG_OK.  (flate2's MultiGzDecoder reports "invalid gzip header" here.)
    bool begin_member() {
        Member m;
        if (at >= n) { done = true; return false; }
        if (!parse_header(in, n, at, m)) {
            bool zeros = at != 0;
            for (size_t i = at; zeros && i < n; ++i) zeros = in[i] == 0;
            if (!zeros) return fail("damaged gzip header");
            done = true; return false;
        }
        inf->start(in, n, m.data);
        in_member = true; crc = 0; produced = 0;
        lo = wr;                                                       // a member has no history
        mhist = 0;
        return true;
    }
    bool end_member() {
        inf->byte_align();
        size_t q = inf->ip;
        if (q + 8 > n) return fail("truncated gzip trailer");
        const u32 want_crc = in[q] | (u32)in[q + 1] << 8 | (u32)in[q + 2] << 16 | (u32)in[q + 3] << 24;
        const u32 want_len = in[q + 4] | (u32)in[q + 5] << 8 | (u32)in[q + 6] << 16 | (u32)in[q + 7] << 24;
        if (want_crc != crc) return fail("gzip CRC mismatch");
        if (want_len != (u32)produced) return fail("gzip length mismatch");
        at = q + 8; in_member = false;
        return true;
    }
    // more bytes into [wr, ..): false when nothing more comes (end or error)
    // keep the history (at most 32 KiB) at the front of the window, write behind it
    void slide() {
        const size_t keep_from = std::max(lo, wr > HIST ? wr - HIST : 0), keep = wr - keep_from;
        if (keep_from) memmove(win.data(), win.data() + keep_from, keep);
        lo = 0; rd = wr = keep;
    }
    bool produce() {
        if (rd != wr) return true;
        if (wr > HIST) slide();
        if (is_bgzf && threads > 1 && !in_member) return produce_bgzf();
        for (;;) {
            if (!in_member && !begin_member()) return false;
            size_t op = wr;
            if (!inf->run(win.data(), lo, op, HIST + CHUNK)) return fail(inf->err);
            if (op > wr) { crc = crc32(crc, win.data() + wr, op - wr); produced += op - wr; }
            const bool got = op > wr;
            wr = op;
            if (inf->st == Inflater::END && !inf->pend_len && !end_member()) return false;
            if (got) return true;
            if (in_member) return fail("deflate stream made no progress");      // (chunk room is never zero here)
        }
    }
    // The same decoding STRAIGHT into a buffer of the caller's (no window, no copy out of it): base[pos .. end) is filled; base[pos - HIST .. pos) must hold the bytes
    // delivered last (the caller carries them from buffer to buffer: GzAhead).  false: nothing more comes (end of the data, or an error: `bad`).  The read-ahead thread
    // spent a tenth of its time copying every inflated byte out of the window (profiles/r06_notes.md).
    bool fill(u8* base, size_t& pos, size_t end) {
        while (pos < end) {
            if (!in_member && !begin_member()) return false;
            size_t op = pos;
            if (!inf->run(base, pos - mhist, op, end)) return fail(inf->err);
            const bool got = op > pos;
            if (got) { crc = crc32(crc, base + pos, op - pos); produced += op - pos; mhist = std::min<size_t>(HIST, mhist + (op - pos)); }
            pos = op;
            if (inf->st == Inflater::END && !inf->pend_len && !end_member()) return false;
            if (!got && in_member) return fail("deflate stream made no progress");
        }
        return true;
    }
    // BGZF: groups of blocks, every block inflated on its own into its place by a standing crew of threads (round 6; until then every group of ~1 MB per thread spawned and
    // joined its threads, and the group was inflated into the window and copied out of it by the one thread that reads ahead: 2.1 - 2.8 GB/s of text however many threads)
    std::vector<Inflater*> pool;
    struct Blk { size_t data, end; u32 isize, crc; size_t out; };
    struct Crew {
        std::vector<std::thread> th; std::mutex m; std::condition_variable go, fin;
        unsigned long long gen = 0; int active = 0, pending = 0; bool quit = false;
        const std::function<void(int)>* job = nullptr;
        void ensure(int n) {                                           // helpers 1 .. n - 1 (the caller is number 0)
            while ((int)th.size() + 1 < n) {
                const int id = (int)th.size() + 1;
                th.emplace_back([this, id]() {
                    unsigned long long seen = 0;
                    for (;;) {
                        const std::function<void(int)>* f = nullptr;
                        { std::unique_lock<std::mutex> g(m); go.wait(g, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; if (id < active) f = job; }
                        if (f) { (*f)(id); std::lock_guard<std::mutex> g(m); if (--pending == 0) fin.notify_all(); }
                    }
                });
            }
        }
        void run(int n, const std::function<void(int)>& f) {
            if (n <= 1) { f(0); return; }
            ensure(n);
            { std::lock_guard<std::mutex> g(m); job = &f; active = n; pending = n - 1; ++gen; }
            go.notify_all();
            f(0);
            std::unique_lock<std::mutex> g(m); fin.wait(g, [&] { return pending == 0; });
        }
        void stop() { { std::lock_guard<std::mutex> g(m); quit = true; } go.notify_all(); for (auto& t : th) t.join(); th.clear(); quit = false; }
        ~Crew() { stop(); }
    } crew;
    // the BGZF blocks from `at` on whose text fits `room` bytes (out: offsets from 0): q <- the position behind them
    bool collect_bgzf(size_t room, std::vector<Blk>& blks, size_t& total, size_t& q) {
        blks.clear(); total = 0; q = at;
        while (q < n) {
            Member m;
            if (!parse_header(in, n, q, m)) break;
            if (!m.bgzf || q + m.bsize > n || m.bsize < (m.data - q) + 8) break;          // an ordinary member: the sequential path takes over from here
            const size_t e = q + m.bsize;
            const u32 c = in[e - 8] | (u32)in[e - 7] << 8 | (u32)in[e - 6] << 16 | (u32)in[e - 5] << 24;
            const u32 isz = in[e - 4] | (u32)in[e - 3] << 8 | (u32)in[e - 2] << 16 | (u32)in[e - 1] << 24;
            if (isz > 65536) return fail("BGZF block larger than 64 KiB");
            if (total + isz > room) break;
            blks.push_back(Blk{m.data, e - 8, isz, c, total});
            total += isz; q = e;
        }
        return true;
    }
    bool inflate_blocks(const std::vector<Blk>& blks, u8* base) {
        const int T = (int)std::min<size_t>((size_t)threads, blks.size());
        while ((int)pool.size() < T) pool.push_back(new Inflater());
        std::vector<const char*> errs((size_t)std::max(T, 1), nullptr);
        const std::function<void(int)> work = [&](int t) {
            Inflater& f = *pool[(size_t)t];
            for (size_t b = (size_t)t; b < blks.size(); b += (size_t)T) {
                const Blk& k = blks[b];
                f.start(in, k.end, k.data);                           // the block's own trailer is the end of its input
                size_t op = k.out;
                if (!f.run(base, k.out, op, k.out + k.isize)) { errs[(size_t)t] = f.err; return; }
                if (f.st != Inflater::END || f.pend_len || op != k.out + k.isize) { errs[(size_t)t] = "BGZF block length mismatch"; return; }
                if (crc32(0, base + k.out, k.isize) != k.crc) { errs[(size_t)t] = "gzip CRC mismatch"; return; }
            }
        };
        crew.run(T, work);
        for (const char* e : errs) if (e) return fail(e);
        return true;
    }
    bool produce_bgzf() {
        std::vector<Blk> blks; size_t total = 0, q = at;
        if (!collect_bgzf(CHUNK * (size_t)threads, blks, total, q)) return false;
        if (blks.empty()) {                                            // not a BGZF block here (or the end): one member the ordinary way
            if (q >= n || !begin_member()) { if (!bad) done = true; return false; }
            is_bgzf = false;
            return produce();
        }
        rd = wr = 0; lo = 0;
        if (win.size() < total + 64) win.resize(total + 64);
        if (!inflate_blocks(blks, win.data())) return false;
        at = q; wr = total;
        if (total == 0) return produce();                              // only empty blocks (the BGZF end marker)
        return true;
    }
    // BGZF blocks STRAIGHT into a buffer of the caller's: as many whole blocks as fit [0, room) (room >= 64 KiB).  Returns the bytes written; 0 with `plain` set: the
    // next member is an ordinary one (or the data ends: `done`), the caller goes on with read(); -1: error
    long fill_bgzf(u8* dst, size_t room, bool& plain) {
        plain = false;
        if (rd != wr || in_member) { plain = true; return 0; }        // (bytes of an ordinary member are still waiting in the window)
        size_t got = 0;
        while (got + 65536 <= room) {
            std::vector<Blk> blks; size_t total = 0, q = at;
            if (!collect_bgzf(room - got, blks, total, q)) return -1;
            if (blks.empty()) { if (at >= n) done = true; else plain = true; break; }
            if (!inflate_blocks(blks, dst + got)) return -1;
            at = q; got += total;
            if (at >= n) { done = true; break; }
        }
        return (long)got;
    }
    void close_pool() { crew.stop(); for (Inflater* f : pool) delete f; pool.clear(); }

    int read(u8* dst, size_t want) {                                   // bytes delivered, 0 at the end, -1 on a malformed stream
        size_t got = 0;
        while (got < want) {
            if (rd != wr) {
                const size_t take = std::min(want - got, wr - rd);
                memcpy(dst + got, win.data() + rd, take); rd += take; got += take;
            } else if (done || bad || !produce()) break;
        }
        return bad ? -1 : (int)got;
    }
};

// The same stream read AHEAD by a thread of its own: the caller parses piece i while piece i + 1 is being inflated (an ordinary gzip file is one
// sequential stream, so this overlap is all the parallelism it offers; a BGZF file is also inflated by GzIn's group threads underneath).
struct GzAhead {
    GzIn core;
    int SLOTS = 3; size_t PIECE = 4u << 20;                                        // how far ahead: set_depth before the first read
    struct Slot { std::unique_ptr<u8[]> data; size_t n = 0; int state = 0; };           // data: HIST bytes of headroom (the history the decoder looks back into), then the piece;
                                                                                        // state: 0 free, 1 filled, 2 last (n bytes, then the end), 3 error
    std::vector<Slot> slot;
    void set_depth(int slots, size_t piece) { if (!started) { SLOTS = std::max(2, slots); PIECE = std::max<size_t>(piece, 1u << 16); } }
    std::mutex mu; std::condition_variable cv;
    std::thread worker; bool stop = false, started = false, direct = true; size_t n_filled = 0;      // direct: decided once, when the worker starts
    int cur = 0; size_t cur_rd = 0; bool finished = false, failed = false;         // consumer side
    ~GzAhead() {
        if (started) { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); worker.join(); }
        core.close_pool();
    }
    void start() {
        started = true;
        direct = !(core.is_bgzf && core.threads > 1);
        slot.resize((size_t)SLOTS);
        worker = std::thread([this]() {
            for (int w = 0;; w = (w + 1) % SLOTS) {
                Slot& s = slot[(size_t)w];
                { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return stop || s.state == 0; }); if (stop) return; }
                constexpr size_t H = GzIn::HIST;
                if (!s.data) s.data.reset(new u8[H + PIECE]);                   // (not zero-filled: touched when written)
                int r; bool last_piece = false, have_last = false;
                if (!direct) {
                    // BGZF: whole blocks inflated by the crew straight into the slot; where an ordinary member interrupts them (or bytes of one are still in the
                    // window) the rest of the piece comes through read()
                    size_t filled = 0; bool plain = false;
                    long g = core.fill_bgzf(s.data.get() + H, PIECE, plain);
                    if (g < 0) r = -1;
                    else {
                        filled = (size_t)g;
                        if (plain) { const int x = core.read(s.data.get() + H + filled, PIECE - filled); if (x < 0) filled = (size_t)-1; else { filled += (size_t)x; last_piece = filled < PIECE; have_last = true; } }
                        else { last_piece = core.done; have_last = true; }
                        r = filled == (size_t)-1 ? -1 : (int)filled;
                    }
                    if (core.bad) r = -1;
                } else {
                    // an ordinary stream: decoded straight into the slot; the 32 KiB in front of the piece are the end of the piece before
                    if (n_filled) { const Slot& pv = slot[(size_t)((w + SLOTS - 1) % SLOTS)]; memcpy(s.data.get(), pv.data.get() + pv.n, H); }
                    size_t pos = H;
                    (void)core.fill(s.data.get(), pos, H + PIECE);
                    r = core.bad ? -1 : (int)(pos - H);
                }
                ++n_filled;
                const bool last = have_last ? last_piece : (size_t)std::max(r, 0) < PIECE;      // (an ordinary stream: a short piece is the last one)
                { std::lock_guard<std::mutex> g(mu); s.n = r > 0 ? (size_t)r : 0; s.state = r < 0 ? 3 : last ? 2 : 1; }
                cv.notify_all();
                if (r < 0 || last) return;
            }
        });
    }
    // the unread bytes of the piece the consumer stands in (waits for the piece): n = 0 at the end of the data; false: the stream is damaged.  consume(k <= n) moves on.
    // (A caller with threads of its own copies a piece out with all of them: read() below is one memcpy on one thread, ~10 GB/s, between rounds of parallel work.)
    bool peek(const u8*& p, size_t& n) {
        if (!started) start();
        for (;;) {
            p = nullptr; n = 0;
            if (failed) return false;
            if (finished) return true;
            Slot& s = slot[(size_t)cur];
            int st;
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return s.state != 0; }); st = s.state; }
            if (st == 3) { failed = true; return false; }
            if (cur_rd < s.n) { p = s.data.get() + GzIn::HIST + cur_rd; n = s.n - cur_rd; return true; }
            if (st == 2) { finished = true; return true; }
            { std::lock_guard<std::mutex> g(mu); s.state = 0; }      // an empty piece that is not the last: on to the next
            cv.notify_all();
            cur = (cur + 1) % SLOTS; cur_rd = 0;
        }
    }
    void consume(size_t k) {
        Slot& s = slot[(size_t)cur];
        cur_rd += k;
        if (cur_rd >= s.n) {
            int st; { std::lock_guard<std::mutex> g(mu); st = s.state; }
            if (st == 2) { finished = true; return; }
            { std::lock_guard<std::mutex> g(mu); s.state = 0; }
            cv.notify_all();
            cur = (cur + 1) % SLOTS; cur_rd = 0;
        }
    }
    int read(u8* dst, size_t want) {
        size_t got = 0;
        while (got < want) {
            const u8* p; size_t n;
            if (!peek(p, n)) return -1;
            if (!n) break;
            const size_t take = std::min(want - got, n);
            memcpy(dst + got, p, take); got += take;
            consume(take);
        }
        return (int)got;
    }
};
}  // namespace gz
