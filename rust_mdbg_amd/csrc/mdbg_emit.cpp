// mdbg_emit.cpp — host-side graph emitter (libmdbg_emit.so).  See include/mdbg_emit.h.
//
// Written against rust-mdbg src/main.rs:1006-1121 (edges, presimp, GFA) and :614-630,693-708 (.sequences); no code is
// shared with the CPU oracle.  Data structures differ from the reference (flat arrays + one hash map from a (k-1)-mer
// to the nodes listing it), the emitted multiset of edges is the same.
#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <system_error>
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>

#include "../../include/mdbg_emit.h"
#include "gz_inflate.h"

namespace {

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

struct Span { const u64* p; u32 n; bool rev; };           // a (k-1)-mer view, read forwards or backwards
inline u64 at(const Span& s, u32 i) { return s.rev ? s.p[s.n - 1 - i] : s.p[i]; }
inline bool equal(const Span& a, const Span& b) { for (u32 i = 0; i < a.n; ++i) if (at(a, i) != at(b, i)) return false; return true; }
// KmerVec::normalize().0 of a span: the lexicographically smaller of the span and its reversal (ties: reversal)
inline Span normalized(const Span& s) {
    for (u32 i = 0; i < s.n; ++i) { u64 a = at(s, i), b = at(s, s.n - 1 - i); if (a < b) return s; if (a > b) break; }
    Span r = s; r.rev = !s.rev; return r;
}
inline u64 hash_span(const Span& s) {
    u64 h = 0x9E3779B97F4A7C15ull;
    for (u32 i = 0; i < s.n; ++i) { h ^= at(s, i); h *= 0xff51afd7ed558ccdull; h ^= h >> 29; }
    return h;
}

// ---- XXH32 (for the LZ4 frame header checksum) ------------------------------------------------------
inline u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
u32 xxh32(const u8* p, size_t len, u32 seed) {
    const u32 P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const u8* end = p + len; u32 h;
    auto rd = [](const u8* q) { u32 v; memcpy(&v, q, 4); return v; };
    if (len >= 16) {
        u32 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do { v1 = rotl32(v1 + rd(p) * P2, 13) * P1; p += 4; v2 = rotl32(v2 + rd(p) * P2, 13) * P1; p += 4;
             v3 = rotl32(v3 + rd(p) * P2, 13) * P1; p += 4; v4 = rotl32(v4 + rd(p) * P2, 13) * P1; p += 4; } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else h = seed + P5;
    h += (u32)len;
    while (p + 4 <= end) { h = rotl32(h + rd(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; ++p; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

// the same hash over a stream (content checksum of an LZ4 frame: XXH32 of everything the frame decodes to)
struct Xxh32Stream {
    u32 v[4]; u8 mem[16]; u32 memsize = 0; u64 total = 0; u32 seed = 0;
    void init(u32 sd) { const u32 P1 = 2654435761u, P2 = 2246822519u; seed = sd; v[0] = sd + P1 + P2; v[1] = sd + P2; v[2] = sd; v[3] = sd - P1; memsize = 0; total = 0; }
    static u32 rd(const u8* q) { u32 x; memcpy(&x, q, 4); return x; }
    static u32 round(u32 acc, u32 in) { return rotl32(acc + in * 2246822519u, 13) * 2654435761u; }
    void update(const u8* p, size_t len) {
        total += len;
        if (memsize + len < 16) { memcpy(mem + memsize, p, len); memsize += (u32)len; return; }
        const u8* end = p + len;
        if (memsize) {
            memcpy(mem + memsize, p, 16 - memsize);
            for (int i = 0; i < 4; ++i) v[i] = round(v[i], rd(mem + 4 * i));
            p += 16 - memsize; memsize = 0;
        }
        while (p + 16 <= end) { for (int i = 0; i < 4; ++i) v[i] = round(v[i], rd(p + 4 * i)); p += 16; }
        if (p < end) { memcpy(mem, p, (size_t)(end - p)); memsize = (u32)(end - p); }
    }
    u32 digest() const {
        const u32 P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
        u32 h = total >= 16 ? rotl32(v[0], 1) + rotl32(v[1], 7) + rotl32(v[2], 12) + rotl32(v[3], 18) : seed + P5;
        h += (u32)total;
        const u8* p = mem; const u8* end = mem + memsize;
        while (p + 4 <= end) { h = rotl32(h + rd(p) * P3, 17) * P4; p += 4; }
        while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; ++p; }
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
        return h;
    }
};

// ---- LZ4 block encoder (the reference writes its .sequences files through lzzzz's frame compressor, src/main.rs:32,65) ----------
// Greedy parse over a table of the last position of every 4-byte hash (one probe, matches extended in both directions), with the
// LZ4 block format's end rules (the last match starts at least 12 bytes before the end of the block, the last 5 bytes are literals).
// `table` is scratch kept by the caller.
void lz4_block_encode(const u8* src, size_t n, std::vector<u8>& dst, std::vector<u32>& table) {
    dst.clear();
    auto rd32 = [&](size_t i) { u32 x; memcpy(&x, src + i, 4); return x; };
    auto put_len = [&](size_t v) { while (v >= 255) { dst.push_back(255); v -= 255; } dst.push_back((u8)v); };
    auto sequence = [&](size_t lit0, size_t lit_n, size_t off, size_t ml) {         // ml = 0: the closing literals-only sequence
        const size_t mc = ml ? ml - 4 : 0;
        dst.push_back((u8)((lit_n >= 15 ? 15 : lit_n) << 4 | (mc >= 15 ? 15 : mc)));
        if (lit_n >= 15) put_len(lit_n - 15);
        dst.insert(dst.end(), src + lit0, src + lit0 + lit_n);
        if (ml) { dst.push_back((u8)(off & 0xFF)); dst.push_back((u8)(off >> 8)); if (mc >= 15) put_len(mc - 15); }
    };
    size_t anchor = 0;
    if (n > 12) {
        table.assign(1u << 16, 0u);                                                  // position + 1 of the last occurrence of a hash
        const size_t mflimit = n - 12, matchlimit = n - 5;
        size_t i = 0, misses = 0;
        while (i < mflimit) {
            const u32 seq = rd32(i);
            const u32 h = (seq * 2654435761u) >> 16;
            const size_t cand = table[h];
            table[h] = (u32)(i + 1);
            if (cand && i - (cand - 1) <= 65535 && rd32(cand - 1) == seq) {
                size_t m = cand - 1;
                while (i > anchor && m > 0 && src[i - 1] == src[m - 1]) { --i; --m; }
                size_t ml = 4;
                while (i + ml < matchlimit && src[i + ml] == src[m + ml]) ++ml;
                sequence(anchor, i - anchor, i - m, ml);
                i += ml; anchor = i; misses = 0;
                if (i >= 2 && i - 2 < mflimit) table[(rd32(i - 2) * 2654435761u) >> 16] = (u32)(i - 1);
            } else i += 1 + (misses++ >> 6);                                        // skip faster through incompressible stretches
        }
    }
    sequence(anchor, n - anchor, 0, 0);
}

char switch_base(char c) {                                    // src/utils.rs:10-24
    switch (c) { case 'a': return 't'; case 'c': return 'g'; case 't': return 'a'; case 'g': return 'c'; case 'u': return 'a';
                 case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C'; case 'U': return 'A'; default: return 'N'; }
}

}  // namespace

struct mdbg_emit { std::vector<u32> n1, n2, ov; std::vector<u8> o1, o2; };

struct mdbg_seqfile {
    FILE* f = nullptr; u32 k = 0, l = 0; std::string buf;
    // LZ4 frame, independent blocks of <= 4 MiB, compressed (lz4_block_encode) unless that does not make them smaller
    std::vector<u8> zbuf; std::vector<u32> ztab;
    bool flush_block() {
        size_t off = 0;
        while (off < buf.size()) {
            const u32 n = (u32)std::min<size_t>(buf.size() - off, 4u << 20);
            lz4_block_encode((const u8*)buf.data() + off, n, zbuf, ztab);
            if (zbuf.size() < n) {
                const u32 hdr = (u32)zbuf.size();
                if (fwrite(&hdr, 4, 1, f) != 1 || fwrite(zbuf.data(), 1, zbuf.size(), f) != zbuf.size()) return false;
            } else {
                const u32 hdr = n | 0x80000000u;              // highest bit: block is not compressed
                if (fwrite(&hdr, 4, 1, f) != 1 || fwrite(buf.data() + off, 1, n, f) != n) return false;
            }
            off += n;
        }
        buf.clear();
        return true;
    }
};

extern "C" {

mdbg_emit* mdbg_emit_create(void) { return new mdbg_emit(); }
void mdbg_emit_destroy(mdbg_emit* e) { delete e; }

int mdbg_emit_edges(mdbg_emit* E, const mdbg_nodes* nd, float presimp, mdbg_edges* out) {
    if (!E || !nd || !out || nd->k < 2) return MDBG_E_PARAM;
    if (nd->n && (!nd->keys || !nd->index || !nd->seqlen || !nd->shift)) return MDBG_E_PARAM;      // a table of mdbg_finalize_gfa: the edges of such a run come from mdbg_graph_edges
    const u64 n = nd->n; const u32 k = nd->k, km = k - 1;
    E->n1.clear(); E->n2.clear(); E->ov.clear(); E->o1.clear(); E->o2.clear();
    // km_index (main.rs:1017-1033): every node is listed under its normalized prefix AND under its normalized suffix
    // (twice in the same list when both coincide).  Buckets are keyed by a hash of the (k-1)-mer; each entry remembers
    // which listing it is, and a query compares the full (k-1)-mer, so hash collisions cannot add or drop a listing.
    struct Listing { u32 row; u8 suffix; };
    std::unordered_map<u64, std::vector<Listing>> index;
    index.reserve(n * 2);
    auto key_of = [&](u64 row, bool suffix) { return normalized(Span{nd->keys + row * k + (suffix ? 1 : 0), km, false}); };
    for (u64 i = 0; i < n; ++i) {
        index[hash_span(key_of(i, false))].push_back({(u32)i, 0});
        index[hash_span(key_of(i, true))].push_back({(u32)i, 1});
    }
    struct Pot { u32 row; char o1, o2; };
    std::vector<Pot> pot;
    struct Ed { u32 r1; char o1; u32 r2; char o2; u32 ov; };
    std::vector<Ed> cand;
    std::unordered_set<u64> removed;
    u64 presimp_removed = 0;
    for (u64 i = 0; i < n; ++i) {                              // main.rs:1041
        const u64* k1 = nd->keys + i * k;
        const Span n1_suf{k1 + 1, km, false}, n1_pre{k1, km, false};
        const Span rev1_suf{k1, km, true};                     // suffix of reversed n1 = its prefix read backwards
        const Span qk[2] = {normalized(n1_suf), normalized(n1_pre)};   // key1 = suffix, key2 = prefix (main.rs:1051-1053)
        for (int q = 0; q < 2; ++q) {
            auto it = index.find(hash_span(qk[q]));
            if (it == index.end()) continue;
            pot.clear();
            for (const Listing& li : it->second) {
                if (!equal(key_of(li.row, li.suffix != 0), qk[q])) continue;
                const u64* k2 = nd->keys + (u64)li.row * k;
                const Span n2_pre{k2, km, false}, rev2_pre{k2 + 1, km, true};   // prefix of reversed n2 = its suffix read backwards
                if (equal(n1_suf, n2_pre)) pot.push_back({li.row, '+', '+'});    // main.rs:1062-1075
                if (equal(n1_suf, rev2_pre)) pot.push_back({li.row, '+', '-'});
                if (equal(rev1_suf, n2_pre)) pot.push_back({li.row, '-', '+'});
                if (equal(rev1_suf, rev2_pre)) pot.push_back({li.row, '-', '-'});
            }
            if (pot.empty()) continue;
            u16 amax = 0; for (auto& p : pot) amax = std::max(amax, nd->abundance[p.row]);
            const u16 aref = std::min(amax, nd->abundance[i]);
            for (auto& p : pot) {                              // main.rs:1078-1102
                if (presimp > 0.0f && pot.size() >= 2 && (float)nd->abundance[p.row] < presimp * (float)aref) {
                    ++presimp_removed; removed.insert(((u64)nd->index[i] << 32) | nd->index[p.row]); continue;
                }
                const u16 shift = p.o1 == '+' ? nd->shift[2 * i] : nd->shift[2 * i + 1];
                const u32 ov = std::min((u32)(nd->seqlen[i] - (u32)shift), (u32)(nd->seqlen[p.row] - 1u));
                cand.push_back({(u32)i, p.o1, p.row, p.o2, ov});
            }
        }
    }
    for (auto& e : cand) {                                     // main.rs:1104-1115
        const u64 a = nd->index[e.r1], b = nd->index[e.r2];
        if (presimp > 0.0f && (removed.count((a << 32) | b) || removed.count((b << 32) | a))) continue;
        E->n1.push_back((u32)a); E->o1.push_back((u8)e.o1); E->n2.push_back((u32)b); E->o2.push_back((u8)e.o2); E->ov.push_back(e.ov);
    }
    out->n = E->n1.size(); out->n1 = E->n1.data(); out->o1 = E->o1.data(); out->n2 = E->n2.data(); out->o2 = E->o2.data();
    out->overlap = E->ov.data(); out->presimp_removed = presimp_removed;
    return MDBG_OK;
}

// decimal digits of v appended at p -> new end (the graph files are millions of short lines: fprintf's format parsing is most of their cost)
static inline char* put_u32(char* p, u32 v) {
    char t[10]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
}
int mdbg_emit_write_gfa(const char* path, const mdbg_nodes* nd, const mdbg_edges* ed) {
    if (!path || !nd) return MDBG_E_PARAM;
    if (nd->n && (!nd->index || !nd->seqlen || !nd->abundance)) return MDBG_E_PARAM;      // (a table of mdbg_finalize_gfa holds these three columns and nothing else)
    FILE* f = fopen(path, "wb");
    if (!f) return MDBG_E_IO;
    // The lines are formatted by a few threads, each a contiguous range of lines into a buffer of its own, and written in order (one thread: 31 ms per 465 k nodes +
    // 910 k edges, a tenth of a file -> .gfa run at 25 Gbases/s).  Small graphs stay on the caller's thread.  The work goes in ROUNDS of at most GFA_ROUND lines
    // (first the S lines, then the L lines): the buffers hold one round, not the file (tens of millions of lines would be gigabytes), and an allocation that fails
    // in a formatting thread ends the call with MDBG_E_NOMEM instead of the process.
    const u64 n_s = nd->n, n_l = ed ? ed->n : 0;
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = n_s + n_l < 200000 ? 1 : (int)std::min<u64>(16, std::max<unsigned>(1, hw));
    static const u64 GFA_ROUND = [] { const char* e = getenv("MDBG_GFA_ROUND_LINES"); const u64 v = e ? strtoull(e, nullptr, 10) : 0; return v ? v : (u64)4 << 20; }();      // (the variable: tests)
    std::vector<std::string> part((size_t)T);
    std::atomic<int> bad{0}, nomem{0};
    struct stat fst;
    const int fd = fileno(f);
    const bool positioned = T > 1 && fstat(fd, &fst) == 0 && S_ISREG(fst.st_mode);      // a regular file: every thread writes its part at its place (the copy into the page
                                                                                         // cache was 10 of the 19 ms of this call for 465 k nodes + 910 k edges)
    auto s_line = [&](u64 i, char* p) -> char* {                                        // main.rs:1021  S\t{index}\t*\tLN:i:{seqlen}\tKC:i:{abundance}
        *p++ = 'S'; *p++ = '\t'; p = put_u32(p, nd->index[i]); memcpy(p, "\t*\tLN:i:", 8); p += 8; p = put_u32(p, nd->seqlen[i]); memcpy(p, "\tKC:i:", 6); p += 6;
        p = put_u32(p, (u32)nd->abundance[i]); *p++ = '\n';
        return p;
    };
    auto l_line = [&](u64 i, char* p) -> char* {                                        // main.rs:1095  L\t{n1}\t{o1}\t{n2}\t{o2}\t{overlap}M
        *p++ = 'L'; *p++ = '\t'; p = put_u32(p, ed->n1[i]); *p++ = '\t'; *p++ = (char)ed->o1[i]; *p++ = '\t'; p = put_u32(p, ed->n2[i]); *p++ = '\t'; *p++ = (char)ed->o2[i]; *p++ = '\t';
        p = put_u32(p, ed->overlap[i]); *p++ = 'M'; *p++ = '\n';
        return p;
    };
    auto put = [&](const char* p, size_t n, u64 off) {
        while (n) { const ssize_t w = pwrite(fd, p, n, (off_t)off); if (w < 0 && errno == EINTR) continue; if (w <= 0) { bad.store(1); return; } p += w; n -= (size_t)w; off += (u64)w; }
    };
    u64 file_at = 11;                                                                   // behind the header line
    bool ok = true;
    if (positioned) put("H\tVN:Z:1.0\n", 11, 0); else ok = fwrite("H\tVN:Z:1.0\n", 1, 11, f) == 11;      // main.rs:1011
    auto emit = [&](u64 n_lines, bool l_lines) {
        for (u64 r0 = 0; r0 < n_lines && ok && !bad.load() && !nomem.load(); r0 += GFA_ROUND) {
            const u64 rn = std::min<u64>(GFA_ROUND, n_lines - r0);
            auto fmt = [&](int t) {
                const u64 i0 = r0 + rn * (u64)t / (u64)T, i1 = r0 + rn * (u64)(t + 1) / (u64)T;
                char tmp[96];
                try {
                    std::string& S = part[(size_t)t]; S.clear(); S.reserve((i1 - i0) * 40 + 64);
                    for (u64 i = i0; i < i1; ++i) { char* const e = l_lines ? l_line(i, tmp) : s_line(i, tmp); S.append(tmp, (size_t)(e - tmp)); }
                } catch (const std::bad_alloc&) { nomem.store(1); }
            };
            {
                std::vector<std::thread> th;
                for (int t = 1; t < T; ++t) th.emplace_back(fmt, t);
                fmt(0);
                for (auto& x : th) x.join();
            }
            if (nomem.load()) return;
            if (positioned) {
                std::vector<u64> at((size_t)T + 1, file_at);
                for (int t = 0; t < T; ++t) at[(size_t)t + 1] = at[(size_t)t] + part[(size_t)t].size();
                auto wr = [&](int t) { put(part[(size_t)t].data(), part[(size_t)t].size(), at[(size_t)t]); };
                std::vector<std::thread> th;
                for (int t = 1; t < T; ++t) th.emplace_back(wr, t);
                wr(0);
                for (auto& x : th) x.join();
                file_at = at[(size_t)T];
            } else
                for (int t = 0; ok && t < T; ++t) ok = part[(size_t)t].empty() || fwrite(part[(size_t)t].data(), 1, part[(size_t)t].size(), f) == part[(size_t)t].size();
        }
    };
    try { emit(n_s, false); emit(n_l, true); } catch (const std::bad_alloc&) { nomem.store(1); } catch (const std::system_error&) { nomem.store(1); }      // (a thread that could not be started)
    ok = ok && !ferror(f);                     // a short write (disk full) must not pass for a complete graph
    ok = (fclose(f) == 0) && ok;
    if (nomem.load()) return MDBG_E_NOMEM;
    return ok && !bad.load() ? MDBG_OK : MDBG_E_IO;
}

mdbg_seqfile* mdbg_seqfile_open(const char* path, uint32_t k, uint32_t l, int* err) {
    FILE* f = path ? fopen(path, "wb") : nullptr;
    if (!f) { if (err) *err = path ? MDBG_E_IO : MDBG_E_PARAM; return nullptr; }
    mdbg_seqfile* s = new mdbg_seqfile(); s->f = f; s->k = k; s->l = l;
    // LZ4 frame header: magic, FLG (version 01, independent blocks), BD (4 MiB blocks), header checksum
    u8 hdr[7] = {0x04, 0x22, 0x4D, 0x18, 0x60, 0x70, 0};
    hdr[6] = (u8)((xxh32(hdr + 4, 2, 0) >> 8) & 0xFF);
    if (fwrite(hdr, 1, 7, f) != 7) { fclose(f); delete s; if (err) *err = MDBG_E_IO; return nullptr; }
    char b[256];
    snprintf(b, sizeof b, "# k = %u\n# l = %u\n", k, l); s->buf += b;                               // main.rs:625-628
    s->buf += "# Structure of remaining of the file:\n";
    s->buf += "# [node name]\t[list of minimizers]\t[sequence of node]\t[abundance]\t[origin]\t[shift]\n";
    if (err) *err = MDBG_OK;
    return s;
}

static inline void put_u64s(std::string& out, u64 v) {
    char t[20]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) out.push_back(t[--n]);
}
// the lines of the nodes i with i % n_parts == part whose A-th sighting lies in this batch
static int seqfile_write_part(mdbg_seqfile* s, const mdbg_nodes* nd, uint32_t part, uint32_t n_parts, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t first) {
    if (!s || !nd || !offsets || (n_reads && !bases && offsets[n_reads]) || !n_parts || part >= n_parts) return MDBG_E_PARAM;
    if (nd->n && (!nd->keys || !nd->src_read || !nd->src_start || !nd->src_end || !nd->reversed || !nd->shift_full || !nd->index)) return MDBG_E_PARAM;      // a table of mdbg_finalize_gfa
    const u32 k = nd->k;
    for (u64 i = part; i < nd->n; i += n_parts) {
        const u64 r = nd->src_read[i];
        if (r < first || r >= first + n_reads) continue;
        const u64 ro = offsets[r - first], a = nd->src_start[i], b = nd->src_end[i];
        if (ro + b > offsets[r - first + 1] || a > b) return MDBG_E_PARAM;
        put_u64s(s->buf, nd->index[i]); s->buf += "\t[";                                              // main.rs:702: {index}\t{node:?}\t{seq}\t*\t{origin}\t{shift:?}
        for (u32 j = 0; j < k; ++j) { if (j) s->buf += ", "; put_u64s(s->buf, nd->keys[i * k + j]); }
        s->buf += "]\t";
        if (nd->reversed[i]) for (u64 p = b; p > a; --p) s->buf += switch_base((char)bases[ro + p - 1]);   // utils::revcomp, main.rs:701
        else s->buf.append((const char*)bases + ro + a, b - a);
        s->buf += "\t*\t*\t("; put_u64s(s->buf, nd->shift_full[2 * i]); s->buf += ", "; put_u64s(s->buf, nd->shift_full[2 * i + 1]); s->buf += ")\n";
        if (s->buf.size() >= (4u << 20) && !s->flush_block()) return MDBG_E_IO;
    }
    return MDBG_OK;
}
int mdbg_seqfile_write_batch(mdbg_seqfile* s, const mdbg_nodes* nd, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t first) {
    return seqfile_write_part(s, nd, 0, 1, bases, offsets, n_reads, first);
}
int mdbg_seqfile_write_batch_part(mdbg_seqfile* s, const mdbg_nodes* nd, uint32_t part, uint32_t n_parts, const uint8_t* bases, const uint64_t* offsets,
                                  uint64_t n_reads, uint64_t first) {
    return seqfile_write_part(s, nd, part, n_parts, bases, offsets, n_reads, first);
}

int mdbg_seqfile_close(mdbg_seqfile* s) {
    if (!s) return MDBG_E_PARAM;
    bool ok = s->flush_block();
    const u32 endmark = 0;
    ok = ok && fwrite(&endmark, 4, 1, s->f) == 1;
    ok = ok && !ferror(s->f);
    ok = (fclose(s->f) == 0) && ok;
    delete s;
    return ok ? MDBG_OK : MDBG_E_IO;
}

}  // extern "C"

// ---- LZ4 frame input (src/main.rs:168-172: a ".lz4" file goes through lzzzz's BufReadDecompressor) --------------------
// Streaming decoder of the LZ4 frame format (magic 0x184D2204; linked or independent blocks, stored or compressed blocks,
// skippable and concatenated frames); the header, block and content checksums (XXH32) are verified where the frame carries them.
struct Lz4In {
    FILE* f = nullptr; bool bad = false, done = false, in_frame = false, blk_sum = false, content_sum = false;
    Xxh32Stream content;
    std::vector<u8> out;              // [history (<= 64 KiB) | bytes of the current block]; rd = next byte to hand out
    size_t rd = 0;
    std::vector<u8> blk;
    bool get(void* p, size_t n) { return fread(p, 1, n, f) == n; }
    bool header() {                   // false: clean end of input (or error, see bad)
        for (;;) {
            u8 m[4];
            const size_t got = fread(m, 1, 4, f);
            if (got == 0) return false;
            if (got != 4) { bad = true; return false; }
            const u32 magic = (u32)m[0] | (u32)m[1] << 8 | (u32)m[2] << 16 | (u32)m[3] << 24;
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {          // skippable frame
                u8 z[4]; if (!get(z, 4)) { bad = true; return false; }
                const u32 n = (u32)z[0] | (u32)z[1] << 8 | (u32)z[2] << 16 | (u32)z[3] << 24;
                if (fseek(f, (long)n, SEEK_CUR) != 0) { bad = true; return false; }
                continue;
            }
            if (magic != 0x184D2204u) { bad = true; return false; }
            u8 fb[2]; if (!get(fb, 2)) { bad = true; return false; }
            if ((fb[0] >> 6) != 1) { bad = true; return false; }
            blk_sum = (fb[0] >> 4) & 1; content_sum = (fb[0] >> 2) & 1;
            size_t skip = 1;                                    // header checksum
            if ((fb[0] >> 3) & 1) skip += 8;                    // content size
            if (fb[0] & 1) skip += 4;                           // dictionary id
            u8 desc[16] = {fb[0], fb[1]};                       // descriptor = FLG, BD, [content size], [dictionary id]; then its checksum byte
            if (!get(desc + 2, skip)) { bad = true; return false; }
            if (((xxh32(desc, 2 + skip - 1, 0) >> 8) & 0xFF) != desc[2 + skip - 1]) { bad = true; return false; }
            content.init(0);
            in_frame = true; out.clear(); rd = 0;
            return true;
        }
    }
    bool decode_block(const u8* src, size_t n) {               // appends to out
        size_t i = 0;
        while (i < n) {
            const u8 tok = src[i++];
            size_t lit = tok >> 4;
            if (lit == 15) { u8 b; do { if (i >= n) return false; b = src[i++]; lit += b; } while (b == 255); }
            if (i + lit > n) return false;
            out.insert(out.end(), src + i, src + i + lit); i += lit;
            if (i >= n) break;                                  // the last sequence holds literals only
            if (i + 2 > n) return false;
            const size_t off = (size_t)src[i] | (size_t)src[i + 1] << 8; i += 2;
            size_t ml = (tok & 15u);
            if (ml == 15) { u8 b; do { if (i >= n) return false; b = src[i++]; ml += b; } while (b == 255); }
            ml += 4;
            if (off == 0 || off > out.size()) return false;
            const size_t from = out.size() - off, to = out.size();
            out.resize(to + ml);
            u8* o = out.data();
            if (off >= ml) memcpy(o + to, o + from, ml);                        // the usual case: source and destination apart
            else for (size_t j = 0; j < ml; ++j) o[to + j] = o[from + j];       // a match that overlaps its own output repeats it: byte by byte
        }
        return true;
    }
    bool next_block() {               // false: end of all input (or error)
        for (;;) {
            if (!in_frame && !header()) return false;
            u8 z[4]; if (!get(z, 4)) { bad = true; return false; }
            const u32 w = (u32)z[0] | (u32)z[1] << 8 | (u32)z[2] << 16 | (u32)z[3] << 24;
            if (w == 0) {                                       // end mark
                if (content_sum) {
                    u8 c4[4]; if (!get(c4, 4)) { bad = true; return false; }
                    if (((u32)c4[0] | (u32)c4[1] << 8 | (u32)c4[2] << 16 | (u32)c4[3] << 24) != content.digest()) { bad = true; return false; }
                }
                in_frame = false;
                continue;
            }
            const size_t n = w & 0x7FFFFFFFu;
            if (n > (8u << 20)) { bad = true; return false; }
            blk.resize(n);
            if (!get(blk.data(), n)) { bad = true; return false; }
            if (blk_sum) {
                u8 c4[4]; if (!get(c4, 4)) { bad = true; return false; }
                if (((u32)c4[0] | (u32)c4[1] << 8 | (u32)c4[2] << 16 | (u32)c4[3] << 24) != xxh32(blk.data(), n, 0)) { bad = true; return false; }
            }
            if (rd > (64u << 10)) { const size_t drop = rd - (64u << 10); out.erase(out.begin(), out.begin() + (long)drop); rd -= drop; }   // keep 64 KiB of history
            const size_t before = out.size();
            if (w & 0x80000000u) out.insert(out.end(), blk.begin(), blk.end());
            else if (!decode_block(blk.data(), n)) { bad = true; return false; }
            if (content_sum) content.update(out.data() + before, out.size() - before);
            return true;
        }
    }
    int read(u8* dst, size_t n) {     // like gzread: bytes delivered, 0 at the end, -1 on a malformed stream
        size_t got = 0;
        while (got < n) {
            if (rd == out.size()) { if (done || !next_block()) { done = true; break; } continue; }
            const size_t take = std::min(n - got, out.size() - rd);
            memcpy(dst + got, out.data() + rd, take); rd += take; got += take;
        }
        return bad ? -1 : (int)got;
    }
};

// ---- host ingest -------------------------------------------------------------------------------------

// A pool of worker threads that lives as long as its reader: run(f) executes f(0) .. f(n - 1), f(0) on the caller.  (std::thread per batch and stage: 64 threads x 3
// stages x ~40 us of spawn + join per 256-Mbase batch was a third of the reader's time at 64 threads, profiles/r04_n_file_pipeline.json -> r05_file_pipeline.json.)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
struct WorkerPool {
    // A window of the reader is 2 - 3 ms of work for all threads, so how a round starts and ends matters: the round number and the count of workers still busy are
    // atomics; a worker that has finished spins on the round number for a moment (the next window usually follows at once) before it goes to sleep on the condition
    // variable, and the caller spins on the busy count (it was a second condition variable: 63 woken workers queueing for one mutex twice per round).
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv_go;
    const std::function<void(int)>* job = nullptr; std::atomic<u64> epoch{0}; std::atomic<int> pending{0}, failed{0}; int n = 1; std::atomic<int> sleepers{0}; bool quit = false;
    explicit WorkerPool(int n_) : n(std::max(1, n_)) {
        for (int i = 1; i < n; ++i) th.emplace_back([this, i] {
            u64 seen = 0;
            for (;;) {
                bool got = false;
                for (int spin = 0; spin < 20000 && !got; ++spin) { if (epoch.load(std::memory_order_acquire) != seen) got = true; else cpu_relax(); }
                if (!got) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1, std::memory_order_relaxed);
                    cv_go.wait(lk, [&] { return quit || epoch.load(std::memory_order_acquire) != seen; });
                    sleepers.fetch_sub(1, std::memory_order_relaxed);
                    if (quit) return;
                }
                seen = epoch.load(std::memory_order_acquire);
                try { (*job)(i); } catch (...) { failed.store(1, std::memory_order_relaxed); }      // (std::bad_alloc from a job's containers: the round's caller reports it)
                pending.fetch_sub(1, std::memory_order_acq_rel);
            }
        });
    }
    // false: a job threw (out of memory in one of its containers): the round's results are not to be used
    bool run(const std::function<void(int)>& f) {
        failed.store(0, std::memory_order_relaxed);
        if (n == 1) { try { f(0); } catch (...) { return false; } return true; }
        job = &f; pending.store(n - 1, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu); epoch.fetch_add(1, std::memory_order_release); }      // (under the mutex: a worker between its predicate and its sleep cannot miss it)
        if (sleepers.load(std::memory_order_relaxed)) cv_go.notify_all();
        try { f(0); } catch (...) { failed.store(1, std::memory_order_relaxed); }
        for (u32 spin = 0; pending.load(std::memory_order_acquire) != 0; ++spin) { if (spin < 4096) cpu_relax(); else sched_yield(); }
        return failed.load(std::memory_order_relaxed) == 0;
    }
    ~WorkerPool() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv_go.notify_all(); for (auto& t : th) t.join(); }
};

struct mdbg_reader {
    gzFile f = nullptr; Lz4In* lz = nullptr; bool fasta = false, strip = false, eof = false, io_error = false;
    gz::GzAhead* gzin = nullptr; const u8* gz_map = nullptr; size_t gz_size = 0;   // a gzip file, mapped and inflated by gz_inflate.h (f stays null)
    bool gz_ahead = false;                                    // inflate on a thread of its own, ahead of the parser (mdbg_reader_open_mt)
    // mdbg_reader_open_mt on gzip input: the inflated text is collected in a window [carry of the previous window | new text], the part of it that holds whole
    // records is parsed by `threads` threads like a window of a mapped file (map points at gw then), the rest is carried over
    u8* gw = nullptr; size_t gw_cap = 0, gw_len = 0; bool gw_mode = false, gw_eof = false;
    std::vector<u8> buf; size_t pos = 0, len = 0;            // input window
    const u8* mem = nullptr;                                  // memory mode (a piece of a mapped file): the window is [mem, mem + len), never refilled
    const u8* data() const { return mem ? mem : buf.data(); }
    // whole-file mapping of an uncompressed input read by several threads (mdbg_reader_open_mt); cur = first unread byte
    const u8* map = nullptr; size_t map_size = 0, map_cur = 0; int threads = 1;
    int fd = -1;                                              // the mapped file, kept open: the fast path READS its chunks (pread into a buffer that stays in L2) instead of touching the mapping
    std::vector<std::vector<u8>> chunk_buf;                   // one per worker
    std::atomic<size_t> margin_hint{32u << 10};               // text read behind a chunk's nominal end to find the record start there (grows to what the records of this file need)
    // the batch buffers handed to the caller (big / big2 / pw / pw2) come from here: malloc / free unless mdbg_reader_set_allocator named something else
    // (mdbg_host_alloc of libmdbg_hip.so: buffers the ingest calls page-lock, so the copy to the device is one DMA instead of a staged memcpy)
    void* (*alloc_fn)(size_t) = nullptr; void (*free_fn)(void*) = nullptr;
    void* buf_alloc(size_t n) { return alloc_fn ? alloc_fn(n) : malloc(n); }
    void buf_free(void* p) { if (!p) return; if (free_fn) free_fn(p); else free(p); }
    bool grow_big(size_t bytes) { if (bytes <= big_cap) return true; buf_free(big); big_cap = bytes; big = (u8*)buf_alloc(big_cap); if (!big) big_cap = 0; return big != nullptr; }
    bool grow_pw(size_t words) { if (words <= pw_cap) return true; buf_free(pw); pw_cap = words; pw = (u64*)buf_alloc(pw_cap * 8); if (!pw) pw_cap = 0; return pw != nullptr; }
    u8* big = nullptr; size_t big_cap = 0;                    // batch buffer of the parallel path (not zero-filled like a vector)
    u8* big2 = nullptr; size_t big2_cap = 0; std::vector<u64> offs2;      // ... the previous batch: the parallel path alternates two buffers, so a
                                                              // batch stays valid while the next one is being read (a packer thread can work on it)
    std::vector<std::vector<u8>> piece_bases; std::vector<std::vector<u64>> piece_lens;      // per-thread parse buffers, kept across batches (no fresh page faults)
    // mdbg_reader_next_packed: the batch as 2-bit words + exception list (two alternating sets, like big / big2)
    u64* pw = nullptr; size_t pw_cap = 0; u64* pw2 = nullptr; size_t pw2_cap = 0;
    std::vector<u64> pexc_pos, pexc_pos2; std::vector<u8> pexc_val, pexc_val2;
    std::vector<u8> bases; std::vector<u64> offs;            // current batch
    std::vector<u8> pending; bool have_pending = false;      // a parsed record that did not fit the previous batch
    // fast path of the parallel reader (round 5): records whose sequence is ONE line are located by a scan (no copy) and packed / copied straight from the mapped
    // text by a pool of workers that lives as long as the reader (rounds 2 - 4: two copies per base and three thread spawns per batch)
    struct FastRec { u64 off, len; };
    std::vector<std::vector<FastRec>> fast_recs;
    struct WorkerPool* pool = nullptr;
    bool packed_done = false;                                 // the batch in offs has been packed by the fast path already (mdbg_reader_next_packed)
    // A window of the mapped file is never looked at again once its batch has been copied / packed: its pages go back on a thread of their own while the next window is read
    // (munmap of the whole 7-GB mapping at close was ~40 ms of a 0.28-s run)
    size_t unmap_from = 0; std::thread unmap_th; std::mutex unmap_mu; std::condition_variable unmap_cv; std::vector<std::pair<size_t, size_t>> unmap_q; bool unmap_quit = false;
    void unmap_upto(size_t end) {
        if (!map || gw_mode) return;
        const size_t page = 4096, to = end & ~(page - 1);
        if (to < unmap_from + (64u << 20)) return;
        if (!unmap_th.joinable()) unmap_th = std::thread([this] {
            for (;;) {
                std::pair<size_t, size_t> q;
                { std::unique_lock<std::mutex> lk(unmap_mu); unmap_cv.wait(lk, [&] { return unmap_quit || !unmap_q.empty(); }); if (unmap_q.empty()) return; q = unmap_q.back(); unmap_q.pop_back(); }
                munmap((void*)(map + q.first), q.second);
            }
        });
        { std::lock_guard<std::mutex> lk(unmap_mu); unmap_q.emplace_back(unmap_from, to - unmap_from); }
        unmap_cv.notify_one();
        unmap_from = to;
    }
    void unmap_finish() {
        if (unmap_th.joinable()) { { std::lock_guard<std::mutex> lk(unmap_mu); unmap_quit = true; } unmap_cv.notify_one(); unmap_th.join(); }
        if (map && !gw_mode && map_size > unmap_from) munmap((void*)(map + unmap_from), map_size - unmap_from);
    }
    bool fill() {                                             // more input; false at EOF
        if (eof || mem) return false;
        if (pos > 0) { memmove(buf.data(), buf.data() + pos, len - pos); len -= pos; pos = 0; }
        if (buf.size() - len < (1u << 20)) buf.resize(buf.size() * 2);
        const size_t want = std::min<size_t>(buf.size() - len, 1u << 30);
        const int n = gzin ? (gz_ahead ? gzin->read(buf.data() + len, want) : gzin->core.read(buf.data() + len, want)) : lz ? lz->read(buf.data() + len, want) : gzread(f, buf.data() + len, (unsigned)want);
        if (n < 0) io_error = true;
        if (n <= 0) { eof = true; return false; }
        len += (size_t)n;
        return true;
    }
    // next line [start, end) without its terminator; false at EOF with nothing left
    bool line(size_t& s, size_t& e, bool& had_nl) {
        for (size_t scan = pos;;) {
            const u8* nl = (const u8*)memchr(data() + scan, '\n', len - scan);
            if (nl) { s = pos; e = (size_t)(nl - data()); pos = e + 1; had_nl = true; return true; }
            scan = len;
            const size_t before = pos;
            if (!fill()) { if (pos == len) return false; s = pos; e = len; pos = len; had_nl = false; return true; }
            scan -= before - pos;                             // window was shifted
        }
    }
    // parses one record's sequence into `out`; false at EOF
    bool record(std::vector<u8>& out) { out.clear(); return record_append(out); }
    // ... appended to `out`
    bool record_append(std::vector<u8>& out) {
        const size_t o0 = out.size();
        size_t s, e; bool nl;
        if (fasta) {
            do { if (!line(s, e, nl)) return false; } while (e == s || data()[s] != '>');   // header line
            // seq_io 0.3 RefRecord::seq(): everything between the header's line end and the record's last line end, interior
            // line terminators included, one trailing '\r' trimmed
            bool first = true;
            for (;;) {
                if (pos == len && !fill()) break;
                if (pos < len && data()[pos] == '>') break;
                if (!line(s, e, nl)) break;
                if (!first) out.push_back('\n');
                out.insert(out.end(), data() + s, data() + e);
                first = false;
            }
            if (strip) { size_t w = o0; for (size_t i = o0; i < out.size(); ++i) { const u8 ch = out[i]; if (ch != '\n' && ch != '\r') out[w++] = ch; } out.resize(w); }   // --reference, main.rs:737
            else if (out.size() > o0 && out.back() == '\r') out.pop_back();
            return true;
        }
        do { if (!line(s, e, nl)) return false; } while (e == s);                            // '@' header
        if (!line(s, e, nl)) return false;                                                   // sequence
        size_t ee = e; if (ee > s && data()[ee - 1] == '\r') --ee;
        out.insert(out.end(), data() + s, data() + ee);
        if (!line(s, e, nl)) return true;                                                    // '+'
        line(s, e, nl);                                                                      // qualities
        return true;
    }
};

extern "C" {

mdbg_reader* mdbg_reader_open(const char* path, int strip_newlines, int* err) {
    if (err) *err = MDBG_E_PARAM;
    if (!path) return nullptr;
    const std::string p(path);
    auto ends = [&](const char* suf) { const size_t n = strlen(suf); return p.size() >= n && p.compare(p.size() - n, n, suf) == 0; };
    mdbg_reader* r = new mdbg_reader();
    if (ends(".lz4")) {                                      // src/main.rs:168-172: LZ4 frame
        FILE* fp = fopen(path, "rb");
        if (!fp) { delete r; return nullptr; }
        r->lz = new Lz4In(); r->lz->f = fp;
    } else {
        // a regular file that starts with the gzip magic is mapped and inflated by gz_inflate.h; everything else (plain text, pipes) goes through zlib's
        // gzread, which passes uncompressed input through
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { delete r; return nullptr; }
        struct stat st;
        u8 magic[3] = {0, 0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 18 && pread(fd, magic, 3, 0) == 3 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 8) {
            void* mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (mp != MAP_FAILED) {
                (void)madvise(mp, (size_t)st.st_size, MADV_SEQUENTIAL);
                r->gz_map = (const u8*)mp; r->gz_size = (size_t)st.st_size;
                r->gzin = new gz::GzAhead(); r->gzin->core.open(r->gz_map, r->gz_size, 1);
            }
        }
        close(fd);
        if (!r->gzin) {
            gzFile f = gzopen(path, "rb");
            if (!f) { delete r; return nullptr; }
            gzbuffer(f, 1u << 20);
            r->f = f;
        }
    }
    r->strip = strip_newlines != 0;
    r->fasta = p.find(".fasta.") != std::string::npos || p.find(".fa.") != std::string::npos || ends(".fa") || ends(".fasta");   // main.rs:463
    r->buf.resize(4u << 20);
    r->offs.push_back(0);
    if (err) *err = MDBG_OK;
    return r;
}

// ---- parallel reader for uncompressed files: the file is mapped, a batch is a window of it cut at record starts, split into one piece
// per thread (again at record starts); every piece is parsed by the SAME record() code as the streaming reader (memory mode), the pieces'
// sequences are then copied side by side.  gzip / LZ4 input cannot be split this way: mdbg_reader_open_mt falls back to the streaming reader.
namespace {
size_t line_start_after(const u8* m, size_t n, size_t p) {                 // first line start >= p
    if (p == 0 || p >= n) return p >= n ? n : 0;
    if (m[p - 1] == '\n') return p;
    const u8* nl = (const u8*)memchr(m + p, '\n', n - p);
    return nl ? (size_t)(nl - m) + 1 : n;
}
size_t line_end(const u8* m, size_t n, size_t p) { const u8* nl = p < n ? (const u8*)memchr(m + p, '\n', n - p) : nullptr; return nl ? (size_t)(nl - m) : n; }
// first record start >= p.  FASTA: a line that starts with '>'.  FASTQ (four-line records, as the streaming reader assumes): a line that
// starts with '@' whose third line starts with '+' and whose second and fourth lines have the same length — a quality line that happens
// to start with '@' fails both tests.
// next_record_start on a PIECE of the text: the same walk over [.., n), and hit_end says whether any of its line searches ran into n without finding a terminator — then
// (and only then) the answer may depend on bytes behind n.  A caller that holds the text up to n < the real end reads more and asks again.
size_t next_record_start_in(const u8* m, size_t n, size_t p, bool fasta, bool& hit_end) {
    hit_end = false;
    auto lend = [&](size_t from) -> size_t { const u8* nl = from < n ? (const u8*)memchr(m + from, '\n', n - from) : nullptr; if (!nl) { hit_end = true; return n; } return (size_t)(nl - m); };
    size_t q;
    if (p == 0 || p >= n) { if (p >= n) hit_end = true; q = p >= n ? n : 0; }
    else if (m[p - 1] == '\n') q = p;
    else { const size_t e = lend(p); q = e < n ? e + 1 : n; }
    if (q >= n) hit_end = true;                                      // the line start itself lies at or behind n
    while (q < n) {
        if (fasta) { if (m[q] == '>') return q; }
        else if (m[q] == '@') {
            const size_t e0 = lend(q), s1 = e0 + 1, e1 = lend(s1), s2 = e1 + 1;
            if (s2 < n && m[s2] == '+') {
                const size_t e2 = lend(s2), s3 = e2 + 1, e3 = lend(s3);
                size_t l1 = e1 - s1, l3 = s3 <= n ? e3 - std::min(s3, n) : 0;
                if (l1 && m[e1 - 1] == '\r') --l1;
                if (l3 && e3 <= n && e3 > 0 && m[e3 - 1] == '\r') --l3;
                if (l1 == l3) return q;
            } else if (s2 >= n) hit_end = true;
        }
        q = lend(q) + 1;
    }
    hit_end = true;
    return n;
}
size_t next_record_start(const u8* m, size_t n, size_t p, bool fasta) {
    size_t q = line_start_after(m, n, p);
    while (q < n) {
        if (fasta) { if (m[q] == '>') return q; }
        else if (m[q] == '@') {
            const size_t e0 = line_end(m, n, q), s1 = e0 + 1, e1 = line_end(m, n, s1), s2 = e1 + 1;
            if (s2 < n && m[s2] == '+') {
                const size_t e2 = line_end(m, n, s2), s3 = e2 + 1, e3 = line_end(m, n, s3);
                size_t l1 = e1 - s1, l3 = s3 <= n ? e3 - std::min(s3, n) : 0;
                if (l1 && m[e1 - 1] == '\r') --l1;
                if (l3 && e3 <= n && e3 > 0 && m[e3 - 1] == '\r') --l3;
                if (l1 == l3) return q;
            }
        }
        q = line_end(m, n, q) + 1;
    }
    return n;
}
}  // namespace

// the fast path (defined behind the packer): 1 = the batch is complete (offsets, and the ASCII copy or the packed words), 0 = a record of the window needs the
// general parser (FASTA sequence over several lines, --reference stripping, stray lines), < 0 = error
static int reader_fast_window(mdbg_reader* r, size_t w0, size_t w1, int T, bool ascii_out);
// ascii_out: the sequences are copied side by side into r->big; otherwise they stay in the per-thread pieces (mdbg_reader_next_packed
// packs them from there) and only the offsets are laid out
static int reader_next_parallel(mdbg_reader* r, uint64_t max_bases, bool ascii_out = true) {
    r->offs.assign(1, 0);
    const u8* m = r->map; const size_t n = r->map_size;
    if (r->map_cur >= n) return MDBG_OK;
    const size_t window = r->fasta ? (size_t)max_bases : 2 * (size_t)max_bases;      // file bytes: never more than max_bases bases unless one record is longer
    size_t end = r->map_cur + std::max<size_t>(window, 1) >= n ? n : next_record_start(m, n, r->map_cur + std::max<size_t>(window, 1), r->fasta);
    if (end <= r->map_cur) end = n;
    const int T = std::max(1, r->threads);
    r->packed_done = false;
    {
        static const bool no_fast = getenv("MDBG_READER_NO_FAST") != nullptr;      // (A/B switch and test hook: the general parser on every window)
        // (the fast path is bound by memory, not by threads: 16 - 24 of them reach what this host gives — 56 - 76 Gbases/s —, 64 get a third of that, across two
        // sockets with the page cache on one; profiles/r05_f_reader_threads_before_cap.json.  MDBG_READER_MAX_WORKERS overrides the cap.)
        static const int max_workers = [] { const char* e = getenv("MDBG_READER_MAX_WORKERS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 24; }();
        const int f = no_fast ? 0 : reader_fast_window(r, r->map_cur, end, std::min(T, max_workers), ascii_out);
        if (f < 0) return f;
        if (f == 1) { r->map_cur = end; r->unmap_upto(end); return MDBG_OK; }
    }
    std::vector<size_t> cut(T + 1);
    cut[0] = r->map_cur; cut[T] = end;
    for (int i = 1; i < T; ++i) {
        const size_t guess = r->map_cur + (size_t)((double)(end - r->map_cur) * i / T);
        cut[i] = std::min(end, std::max(cut[i - 1], next_record_start(m, end, guess, r->fasta)));
    }
    r->piece_bases.resize(T); r->piece_lens.resize(T);
    struct PieceRef { std::vector<u8>& bases; std::vector<u64>& lens; };
    std::vector<PieceRef> pc;
    for (int i = 0; i < T; ++i) { r->piece_bases[i].clear(); r->piece_lens[i].clear(); pc.push_back(PieceRef{r->piece_bases[i], r->piece_lens[i]}); }
    auto parse = [&](int i) {
        if (cut[i + 1] <= cut[i]) return;
        mdbg_reader t; t.mem = m + cut[i]; t.len = cut[i + 1] - cut[i]; t.eof = true; t.fasta = r->fasta; t.strip = r->strip;
        pc[i].bases.reserve(t.len / (r->fasta ? 1 : 2) + 64);
        for (size_t before = 0; t.record_append(pc[i].bases); before = pc[i].bases.size()) pc[i].lens.push_back(pc[i].bases.size() - before);
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < T; ++i) th.emplace_back(parse, i);
        parse(0);
        for (auto& x : th) x.join();
    }
    size_t total = 0, reads = 0;
    std::vector<size_t> base0(T), read0(T);
    for (int i = 0; i < T; ++i) { base0[i] = total; read0[i] = reads; total += pc[i].bases.size(); reads += pc[i].lens.size(); }
    if (ascii_out && total + 64 > r->big_cap && !r->grow_big(total + total / 8 + 4096)) return MDBG_E_NOMEM;
    r->offs.resize(reads + 1);
    auto place = [&](int i) {
        if (ascii_out && !pc[i].bases.empty()) memcpy(r->big + base0[i], pc[i].bases.data(), pc[i].bases.size());
        u64 o = base0[i];
        for (size_t j = 0; j < pc[i].lens.size(); ++j) { r->offs[read0[i] + j] = o; o += pc[i].lens[j]; }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < T; ++i) th.emplace_back(place, i);
        place(0);
        for (auto& x : th) x.join();
    }
    r->offs[reads] = total;
    r->map_cur = end;
    r->unmap_upto(end);
    return MDBG_OK;
}

// gzip input read by several threads: [0, E) of the window = the whole records in it.  FASTA: everything in front of the last header line.  FASTQ: up to the
// end of the last record whose four lines verify (next_record_start) — found from the last MB on, from the front if the tail holds none.
namespace {
size_t complete_prefix(const u8* m, size_t n, bool fasta) {
    if (fasta) {
        for (size_t p = n; p > 0;) {
            const u8* g = (const u8*)memrchr(m, '>', p);
            if (!g) return 0;
            const size_t i = (size_t)(g - m);
            if (i == 0 || m[i - 1] == '\n') return i;
            p = i;
        }
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
        size_t p = pass == 0 && n > (1u << 20) ? n - (1u << 20) : 0, last_end = 0;
        bool any = false;
        size_t q = next_record_start(m, n, p, false);
        while (q < n) {
            size_t e = q; bool whole = true;
            for (int l = 0; l < 4 && whole; ++l) { const size_t le = line_end(m, n, e); whole = le < n; e = le + 1; }      // behind the record's fourth line
            if (!whole) break;                                                    // its last line has no terminator in the window yet (a "\r\n" may be cut in two): carried over
            last_end = e; any = true;
            if (last_end >= n) break;
            q = next_record_start(m, n, last_end, false);
        }
        if (any) return last_end;
        if (p == 0) break;
    }
    return 0;
}
}  // namespace
static bool reader_more(const mdbg_reader* r) { return r->gw_mode ? !(r->gw_eof && r->gw_len == 0) : r->map_cur < r->map_size; }
static int reader_next_window(mdbg_reader* r, uint64_t max_bases, bool ascii_out) {
    if (!r->gw_mode) return reader_next_parallel(r, max_bases, ascii_out);
    size_t want = std::max<size_t>(r->fasta ? (size_t)max_bases : 2 * (size_t)max_bases, 1u << 16);
    size_t E = 0;
    for (;;) {
        if (r->gw_cap < want + 64) {
            u8* nb = (u8*)malloc(want + want / 8 + 64);
            if (!nb) return MDBG_E_NOMEM;
            if (r->gw_len) memcpy(nb, r->gw, r->gw_len);
            free(r->gw); r->gw = nb; r->gw_cap = want + want / 8 + 64;
        }
        while (!r->gw_eof && r->gw_len < want) {
            // a piece of inflated text -> the window, copied by all the reader's threads (one memcpy on this thread was the serial stretch between the parallel
            // inflate of BGZF blocks and the parallel parse: 2.3 GB/s of text with 32 threads)
            const u8* p; size_t n;
            if (!r->gzin->peek(p, n)) { r->io_error = true; return MDBG_E_IO; }
            if (n == 0) { r->gw_eof = true; break; }
            const size_t take = std::min(n, want - r->gw_len);
            const int T = std::max(1, r->threads);
            if (T > 1 && take >= (4u << 20)) {
                if (!r->pool || r->pool->n != T) { delete r->pool; r->pool = new WorkerPool(T); }
                u8* const d = r->gw + r->gw_len;
                const size_t per = ((take + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
                const std::function<void(int)> cp = [&](int i) { const size_t a = (size_t)i * per; if (a < take) memcpy(d + a, p + a, std::min(per, take - a)); };
                if (!r->pool->run(cp)) return MDBG_E_NOMEM;
            } else memcpy(r->gw + r->gw_len, p, take);
            r->gzin->consume(take); r->gw_len += take;
        }
        if (r->gw_len == 0) { r->offs.assign(1, 0); return MDBG_OK; }
        E = r->gw_eof ? r->gw_len : complete_prefix(r->gw, r->gw_len, r->fasta);
        if (E > 0) break;
        want *= 2;                                                               // one record is larger than the window: read on
    }
    r->map = r->gw; r->map_size = E; r->map_cur = 0;
    const int e = reader_next_parallel(r, (uint64_t)E, ascii_out);             // the whole of [0, E) is one batch
    r->map = r->gw; r->map_size = 0; r->map_cur = 0;
    memmove(r->gw, r->gw + E, r->gw_len - E); r->gw_len -= E;
    return e;
}

int mdbg_reader_is_fasta(const mdbg_reader* r) { return r && r->fasta ? 1 : 0; }
int mdbg_reader_set_allocator(mdbg_reader* r, void* (*alloc_fn)(size_t), void (*free_fn)(void*)) {
    if (!r || !alloc_fn != !free_fn) return MDBG_E_PARAM;
    if (r->big || r->big2 || r->pw || r->pw2) return MDBG_E_STATE;      // before the first batch: a buffer goes back where it came from
    r->alloc_fn = alloc_fn; r->free_fn = free_fn;
    return MDBG_OK;
}
int mdbg_reader_is_parallel(const mdbg_reader* r) { return r && r->map ? 1 : 0; }

mdbg_reader* mdbg_reader_open_mt(const char* path, int strip_newlines, int threads, int* err) {
    mdbg_reader* r = mdbg_reader_open(path, strip_newlines, err);
    if (!r || threads <= 1 || r->lz) return r;
    // gzip: the inflate runs ahead of the parser on its own thread; BGZF blocks are inflated by the remaining threads (an ordinary gzip file is one stream)
    if (r->gzin) {
        // the thread budget is SPLIT: the read-ahead worker + inflate threads on one side (BGZF blocks; an ordinary stream is decoded by one thread, see
        // gz_inflate.h), the parser threads of a window on the other — together `threads`, not twice that
        // BGZF: inflating a byte costs ~20 x what parsing and packing it does (1 GB/s against 2.7 per thread, and the parser's share shrinks with its window in
        // cache): three threads in four inflate.  An ordinary stream has ONE inflating thread whatever the budget: the rest parse.
        const int inflate = r->gzin->core.is_bgzf ? std::min(48, std::max(1, threads * 3 / 4)) : 1;
        r->gz_ahead = true; r->gzin->core.threads = inflate;
        r->gw_mode = true; r->threads = std::max(1, threads - inflate);
        r->gzin->set_depth(8, 16u << 20);                                        // up to 128 MB of text inflated while the previous window is parsed
        r->gw_cap = 1u << 20; r->gw = (u8*)malloc(r->gw_cap);
        if (!r->gw) { mdbg_reader_close(r); if (err) *err = MDBG_E_NOMEM; return nullptr; }
        r->map = r->gw;                                                          // (non-null: the batch calls take the parallel path)
        return r;
    }
    // plain file?  (gzip magic 1f 8b; ".lz4" was taken by name above)  Map it and let `threads` threads parse it.
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return r;
    struct stat st;
    u8 magic[2] = {0, 0};
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2 || pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { close(fd); return r; }
    void* mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (mp == MAP_FAILED) { close(fd); return r; }
    r->fd = fd;
    // (no MADV_SEQUENTIAL: the fast path reads every window twice — the scan, then the pack / copy — and that advice lets the kernel drop the pages in between)
    r->map = (const u8*)mp; r->map_size = (size_t)st.st_size; r->map_cur = 0; r->threads = threads;
    return r;
}

int mdbg_reader_next(mdbg_reader* r, uint64_t max_bases, const uint8_t** bases, const uint64_t** offsets, uint64_t* n_reads) {
    if (!r || !bases || !offsets || !n_reads) return MDBG_E_PARAM;
    if (r->map) {
        int e;
        std::swap(r->big, r->big2); std::swap(r->big_cap, r->big2_cap); r->offs.swap(r->offs2);       // the batch handed out last stays intact during this call
        do e = reader_next_window(r, max_bases, true); while (!e && r->offs.size() == 1 && reader_more(r));      // a window without a record (text in front of the first header) is not the end
        *bases = r->big; *offsets = r->offs.data(); *n_reads = r->offs.size() - 1;
        return e;
    }
    r->bases.clear(); r->offs.assign(1, 0);
    if (r->have_pending) { r->bases.insert(r->bases.end(), r->pending.begin(), r->pending.end()); r->offs.push_back(r->bases.size()); r->have_pending = false; }
    while (r->bases.size() < max_bases || r->offs.size() == 1) {
        const size_t before = r->bases.size();
        if (!r->record_append(r->bases)) break;                 // (parsed straight into the batch: one copy less per base than through a record buffer)
        if (r->offs.size() > 1 && r->bases.size() > max_bases) {  // does not fit any more: it opens the next batch
            r->pending.assign(r->bases.begin() + (long)before, r->bases.end()); r->bases.resize(before); r->have_pending = true;
            break;
        }
        r->offs.push_back(r->bases.size());
    }
    *bases = r->bases.data(); *offsets = r->offs.data(); *n_reads = r->offs.size() - 1;
    return r->io_error ? MDBG_E_IO : MDBG_OK;                // a malformed / truncated compressed stream
}

void mdbg_reader_close(mdbg_reader* r) {
    if (!r) return;
    static const bool timing = getenv("MDBG_READER_TIMING") != nullptr;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t0 = timing ? now() : 0;
    delete r->pool;
    const double t1 = timing ? now() : 0;
    if (r->fd >= 0) close(r->fd);
    if (r->gzin) delete r->gzin;
    if (r->gz_map) munmap((void*)r->gz_map, r->gz_size);
    r->unmap_finish();
    const double t2 = timing ? now() : 0;
    free(r->gw); r->buf_free(r->big); r->buf_free(r->big2); r->buf_free(r->pw); r->buf_free(r->pw2);
    const double t3 = timing ? now() : 0;
    if (r->f) gzclose(r->f);
    if (r->lz) { if (r->lz->f) fclose(r->lz->f); delete r->lz; }
    delete r;
    if (timing) fprintf(stderr, "[mdbg reader] close: workers %.2f ms, unmap %.2f, batch buffers %.2f, the rest %.2f\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
}

}  // extern "C"

// ---- host packer: ASCII -> two 32-bit planes per 32 bases (mdbg_packed_batch of mdbg_hip.h) -------------------------
#include <immintrin.h>
#include <thread>
namespace {
// planes of 32 bases at p (n <= 32 valid): bit i of lo / hi = bit 1 / bit 2 of byte i; *bad = mask of bytes outside ACGT
inline void pack32_scalar(const uint8_t* p, unsigned n, uint32_t& lo, uint32_t& hi, uint32_t& bad) {
    lo = hi = bad = 0;
    for (unsigned i = 0; i < n; ++i) {
        const uint8_t c = p[i];
        lo |= (uint32_t)((c >> 1) & 1u) << i; hi |= (uint32_t)((c >> 2) & 1u) << i;
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') bad |= 1u << i;
    }
}
__attribute__((target("avx2"))) inline void pack32_avx2(const uint8_t* p, uint32_t& lo, uint32_t& hi, uint32_t& bad) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)p);
    lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 6));      // bit 1 of every byte -> its sign bit
    hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 5));      // bit 2
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(v, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(v, _mm256_set1_epi8('T'))));
    bad = ~(uint32_t)_mm256_movemask_epi8(ok);
}
inline void pack32_sse2(const uint8_t* p, uint32_t& lo, uint32_t& hi, uint32_t& bad) {
    lo = hi = 0; uint32_t okm = 0;
    for (int h = 0; h < 2; ++h) {
        const __m128i v = _mm_loadu_si128((const __m128i*)(p + 16 * h));
        lo |= (uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 6)) << (16 * h);
        hi |= (uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 5)) << (16 * h);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('A')), _mm_cmpeq_epi8(v, _mm_set1_epi8('C'))),
                                        _mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('G')), _mm_cmpeq_epi8(v, _mm_set1_epi8('T'))));
        okm |= (uint32_t)_mm_movemask_epi8(ok) << (16 * h);
    }
    bad = ~okm;
}
struct ExcList { std::vector<uint64_t> pos; std::vector<uint8_t> val; };
void pack_range(const uint8_t* bases, uint64_t n_bases, uint64_t w0, uint64_t w1, uint64_t* words, ExcList& ex, bool avx2) {
    for (uint64_t w = w0; w < w1; ++w) {
        const uint64_t p = w * 32;
        uint32_t lo, hi, bad;
        if (p + 32 <= n_bases) { if (avx2) pack32_avx2(bases + p, lo, hi, bad); else pack32_sse2(bases + p, lo, hi, bad); }
        else pack32_scalar(bases + p, (unsigned)(n_bases - p), lo, hi, bad);
        words[w] = (uint64_t)lo | ((uint64_t)hi << 32);
        while (bad) { const unsigned i = (unsigned)__builtin_ctz(bad); bad &= bad - 1; ex.pos.push_back(p + i); ex.val.push_back(bases[p + i]); }
    }
}
}  // namespace

// Reader batches straight to the packed layout: every parser thread packs the sequences of its own piece (they are contiguous in its
// buffer) into the words that lie entirely inside the piece's range of the batch; the few words that straddle two pieces are assembled
// from per-thread partial contributions afterwards.  No ASCII copy of the batch is written.
namespace {
struct Partial { u64 word; u32 lo, hi; };
void pack_piece(const u8* src, u64 B, u64 E, u64* words, ExcList& ex, std::vector<Partial>& parts, bool avx2) {
    if (E <= B) return;
    auto partial = [&](u64 a, u64 b) {                       // bases [a, b) inside one word
        u32 lo, hi, bad;
        pack32_scalar(src + (a - B), (unsigned)(b - a), lo, hi, bad);
        const unsigned sh = (unsigned)(a & 31);
        parts.push_back(Partial{a >> 5, lo << sh, hi << sh});
        while (bad) { const unsigned i = (unsigned)__builtin_ctz(bad); bad &= bad - 1; ex.pos.push_back(a + i); ex.val.push_back(src[a - B + i]); }
    };
    const u64 w_lo = (B + 31) >> 5, w_hi = E >> 5;           // full words [w_lo, w_hi)
    if (w_lo > w_hi) { partial(B, E); return; }              // the piece lies inside one word
    if (B & 31) partial(B, w_lo << 5);
    for (u64 w = w_lo; w < w_hi; ++w) {
        const u8* p = src + ((w << 5) - B);
        u32 lo, hi, bad;
        if (avx2) pack32_avx2(p, lo, hi, bad); else pack32_sse2(p, lo, hi, bad);
        words[w] = (u64)lo | ((u64)hi << 32);
        while (bad) { const unsigned i = (unsigned)__builtin_ctz(bad); bad &= bad - 1; ex.pos.push_back((w << 5) + i); ex.val.push_back(p[i]); }
    }
    if (E & 31) partial(w_hi << 5, E);
}
int reader_pack_pieces(mdbg_reader* r, u64 total) {
    const size_t T = r->piece_bases.size();
    const u64 nw = (total + 31) / 32;
    if (nw + 8 > r->pw_cap && !r->grow_pw(nw + nw / 8 + 64)) return MDBG_E_NOMEM;
    const bool avx2 = __builtin_cpu_supports("avx2");
    std::vector<u64> base0(T + 1, 0);
    for (size_t i = 0; i < T; ++i) base0[i + 1] = base0[i] + r->piece_bases[i].size();
    std::vector<ExcList> ex(T); std::vector<std::vector<Partial>> parts(T);
    auto work = [&](size_t i) { pack_piece(r->piece_bases[i].data(), base0[i], base0[i + 1], r->pw, ex[i], parts[i], avx2); };
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < T; ++i) th.emplace_back(work, i);
        if (T) work(0);
        for (auto& x : th) x.join();
    }
    for (size_t i = 0; i < T; ++i) for (const Partial& q : parts[i]) r->pw[q.word] = 0;       // words shared by pieces: clear, then OR the contributions
    for (size_t i = 0; i < T; ++i) for (const Partial& q : parts[i]) r->pw[q.word] |= (u64)q.lo | ((u64)q.hi << 32);
    r->pexc_pos.clear(); r->pexc_val.clear();
    for (size_t i = 0; i < T; ++i) { r->pexc_pos.insert(r->pexc_pos.end(), ex[i].pos.begin(), ex[i].pos.end()); r->pexc_val.insert(r->pexc_val.end(), ex[i].val.begin(), ex[i].val.end()); }
    return MDBG_OK;
}
}  // namespace

// ---- fast path of the parallel reader -------------------------------------------------------------------------------------------------------
// Stage A (every worker on its piece of the window, which starts at a record start): the records' sequences are LOCATED — (offset in the text, length) —
// without copying anything; a record whose sequence is not one line of the text sends the whole window to the general parser.  Stage B: the
// offsets are laid out and every worker copies (ASCII batches) or packs (2-bit batches) the sequences of its own records straight from the text.
namespace {
// false: the piece holds something the general parser has to look at
bool scan_piece(const u8* m, size_t a, size_t b, size_t file_end, bool fasta, bool strip, std::vector<mdbg_reader::FastRec>& recs, u64& n_bases) {
    recs.clear(); n_bases = 0;
    size_t p = a;
    auto eol = [&](size_t from) -> size_t { const u8* nl = from < b ? (const u8*)memchr(m + from, '\n', b - from) : nullptr; return nl ? (size_t)(nl - m) : b; };
    while (p < b) {
        if (fasta) {
            if (m[p] != '>') return false;                                  // a stray line in front of a header: the general parser skips it
            const size_t he = eol(p);
            if (he >= b) { recs.push_back({b, 0}); break; }                   // a header without a line behind it: an empty record
            size_t s = he + 1, e = eol(s);
            if (s < b && m[s] == '>') { recs.push_back({s, 0}); p = s; continue; }      // empty sequence
            if (strip) return false;                                        // --reference: line terminators are taken out of the sequence (a copy)
            const size_t next = e < b ? e + 1 : b;
            if (next < b && m[next] != '>') return false;                   // the sequence goes on in the next line: interior terminators belong to it
            if (e == b && b < file_end) return false;                       // (cannot happen: pieces end at record starts)
            size_t ee = e; if (ee > s && m[ee - 1] == '\r') --ee;
            recs.push_back({s, ee - s}); n_bases += ee - s;
            p = next;
        } else {
            size_t he = eol(p);
            if (he == p) { p = he + 1; continue; }                           // the general parser skips empty lines in front of a header
            if (he >= b) break;                                              // a header and nothing else: no record (as the general parser: its sequence line is missing)
            const size_t s = he + 1;
            if (s >= b) break;                                               // "@header\n" and the end of the text: the same — no sequence line, no record (record_append)
            const size_t e = eol(s);
            size_t ee = e; if (ee > s && m[ee - 1] == '\r') --ee;
            recs.push_back({s, ee - s}); n_bases += ee - s;
            if (e >= b) break;
            const size_t pe = eol(e + 1);                                    // '+'
            if (pe >= b) break;
            const size_t qe = eol(pe + 1);                                   // qualities
            p = qe < b ? qe + 1 : b;
        }
    }
    return true;
}
}  // namespace
static int reader_fast_window(mdbg_reader* r, const size_t w0, const size_t w1, const int T, bool ascii_out) {
    // One pass over the text.  The window is taken in CHUNKS of ~256 KB (cut at record starts) that the workers claim in file order; a worker locates the records of its
    // chunk (stage A), waits until the chunk in front has published where its own records go in the batch (a running sum of bases and records handed down the
    // chunks: the workers in front were started earlier on equally sized chunks, so the wait is short), publishes the sum for the next chunk, and packs / copies
    // its records (stage B) while the chunk's text is still in its L2.  (The first version of this path ran stage A over the whole window, then stage B over the
    // whole window: every byte came from DRAM twice.)  The batch buffer is sized BEFORE the scan from a bound: a base is at least one byte of a FASTA text,
    // two of a FASTQ text (a record that breaks the bound sends the window to the general parser).
    // A chunk's text is READ (pread into the worker's buffer), not touched through the mapping: 16 threads walking a mapping of page-cache pages take a minor fault
    // per 16 pages and get 53 - 62 GB/s out of this host, the same threads reading into 256-KB buffers 210 - 244 (scratch/ubench/mmap_vs_pread.cpp,
    // profiles/r05_f_mmap_vs_pread.txt); the mapping is still what the chunk boundaries are looked up in (a few pages per chunk) and what the general parser walks.
    const u8* m = r->map;
    const bool by_read = r->fd >= 0 && !r->gw_mode;
    if (r->chunk_buf.size() < (size_t)T) r->chunk_buf.resize((size_t)T);
    if (!r->pool || r->pool->n != T) { delete r->pool; r->pool = new WorkerPool(T); }
    static const size_t chunk_bytes = [] { const char* e = getenv("MDBG_READER_CHUNK_BYTES"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v : (size_t)256 << 10; }();      // (tests: chunks of a few bytes)
    const size_t C = std::max<size_t>(1, (w1 - w0 + chunk_bytes - 1) / chunk_bytes);
    const u64 cap_bases = r->fasta ? (u64)(w1 - w0) : (u64)(w1 - w0) / 2 + 4096;
    if (ascii_out) { if (cap_bases + 64 > r->big_cap && !r->grow_big(cap_bases + cap_bases / 64 + (64u << 10))) return MDBG_E_NOMEM; }
    else { const u64 nwb = (cap_bases + 31) / 32; if (nwb + 8 > r->pw_cap && !r->grow_pw(nwb + nwb / 64 + 1024)) return MDBG_E_NOMEM; }
    if (r->fast_recs.size() < C) r->fast_recs.resize(C);
    // one cache line per link of the chain (flag + the two sums): the workers behind spin on THEIR link only, a publication moves one line once
    struct alignas(64) Link { std::atomic<u32> ready; u64 base0, read0; };
    std::unique_ptr<Link[]> link(new Link[C + 1]);
    for (size_t c = 0; c <= C; ++c) { link[c].ready.store(c == 0 ? 1u : 0u, std::memory_order_relaxed); link[c].base0 = link[c].read0 = 0; }
    std::atomic<size_t> next_chunk{0};
    std::atomic<int> bad{0}, io_bad{0}, nomem{0};
    const bool avx2 = __builtin_cpu_supports("avx2");
    std::vector<ExcList> ex((size_t)T); std::vector<std::vector<Partial>> parts((size_t)T);
    static const bool timing = getenv("MDBG_READER_TIMING") != nullptr;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t0 = timing ? now() : 0;
    std::vector<double> waited((size_t)T, 0.0);
    auto work_body = [&](int i) {
        for (;;) {
            const size_t c = next_chunk.fetch_add(1, std::memory_order_relaxed);
            if (c >= C || bad.load(std::memory_order_relaxed)) return;
            // the chunk's records: those that start in [n0, n1) of the text, i.e. [a, b) with a / b = the first record start at or behind n0 / n1 (as next_record_start
            // on the whole window finds them: the chunk in front computes the same b as this one's a)
            const size_t n0 = w0 + c * chunk_bytes, n1 = std::min(w1, n0 + chunk_bytes);
            size_t a, b;
            std::vector<mdbg_reader::FastRec>& recs = r->fast_recs[c];
            u64 nb = 0;
            const u8* mc = m;                                      // base pointer under which this chunk's bytes [a, b) are found at their file offsets
            if (!by_read) {
                a = c == 0 ? w0 : next_record_start(m, w1, n0, r->fasta); b = c + 1 == C ? w1 : next_record_start(m, w1, n1, r->fasta);
            } else {
                // ... looked up in the worker's own copy of the text: [n0 - 1, n1 + margin), read on until neither answer can depend on what lies behind the copy.
                // (Through the mapping these two lookups were ~2,000 minor faults per window, serviced at about one per microsecond whatever the number of threads:
                // most of the 2.7 ms a 268-MB window took with 16 threads.)
                std::vector<u8>& tb = r->chunk_buf[(size_t)i];
                const size_t lo = c == 0 ? w0 : n0 - 1;
                static const size_t margin_test = [] { const char* e = getenv("MDBG_READER_MARGIN_BYTES"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v : (size_t)0; }();      // (tests: margins of a few bytes, so that reading on happens)
                size_t hi = lo, want = std::min(w1, n1 + (margin_test ? margin_test : r->margin_hint.load(std::memory_order_relaxed)));
                for (;;) {
                    if (tb.size() < want - lo) tb.resize(want - lo + (want - lo) / 4);
                    while (hi < want) { const ssize_t n = pread(r->fd, tb.data() + (hi - lo), want - hi, (off_t)hi); if (n <= 0) { if (n < 0 && errno == EINTR) continue; break; } hi += (size_t)n; }
                    if (hi < want) { io_bad.store(1, std::memory_order_relaxed); bad.store(1, std::memory_order_relaxed); return; }      // (the file shrank under the mapping)
                    mc = tb.data() - lo;
                    bool ha = false, hb = false;
                    a = c == 0 ? w0 : next_record_start_in(mc, hi, n0, r->fasta, ha);
                    b = c + 1 == C ? w1 : next_record_start_in(mc, hi, n1, r->fasta, hb);
                    if (hi >= w1 || !(ha || hb)) break;            // (with the whole window in hand the answers are next_record_start's on the window)
                    // records of many megabytes (a chromosome on one line): every chunk inside one would read to its end — the general parser cuts a window into
                    // one piece per thread and looks T boundaries up, not thousands
                    if (hi > n1 && hi - n1 > (8u << 20)) { bad.store(1, std::memory_order_relaxed); return; }
                    want = std::min(w1, hi + std::max<size_t>(hi > n1 ? hi - n1 : 0, margin_test ? margin_test : (size_t)(32u << 10)));      // (doubling: a long record is a few more reads, not thousands)
                }
                const size_t used = hi > n1 ? hi - n1 : 0;          // the next chunks start with the margin this one needed (bounded: one long record must not tax all that follow)
                if (used > r->margin_hint.load(std::memory_order_relaxed) && used <= (1u << 20)) r->margin_hint.store((used + 0x3FFF) & ~(size_t)0x3FFF, std::memory_order_relaxed);
            }
            const bool ok = a >= b ? (recs.clear(), true) : scan_piece(mc, a, b, w1, r->fasta, r->strip, recs, nb);
            const double tw = timing ? now() : 0;
            for (u32 spins = 0; !link[c].ready.load(std::memory_order_acquire); ++spins) {
                if (bad.load(std::memory_order_relaxed)) return;
                if (spins < 4096) cpu_relax(); else sched_yield();
            }
            if (timing) waited[(size_t)i] += now() - tw;
            if (!ok || link[c].base0 + nb > cap_bases) { bad.store(1, std::memory_order_relaxed); return; }
            link[c + 1].base0 = link[c].base0 + nb; link[c + 1].read0 = link[c].read0 + recs.size();
            link[c + 1].ready.store(1, std::memory_order_release);
            u64 o = link[c].base0;
            for (const mdbg_reader::FastRec& q : recs) {
                if (q.len) { if (ascii_out) memcpy(r->big + o, mc + q.off, q.len); else pack_piece(mc + q.off, o, o + q.len, r->pw, ex[(size_t)i], parts[(size_t)i], avx2); }
                o += q.len;
            }
        }
    };
    // (the job allocates — chunk buffers, record and exception lists —: a failure there ends the window with MDBG_E_NOMEM like the allocations of the calling thread; `bad`
    // releases the workers that wait for the failed chunk's link)
    const std::function<void(int)> work = [&](int i) { try { work_body(i); } catch (const std::bad_alloc&) { nomem.store(1, std::memory_order_relaxed); bad.store(1, std::memory_order_relaxed); } };
    if (!r->pool->run(work) || nomem.load()) return MDBG_E_NOMEM;
    if (io_bad.load()) { r->io_error = true; return MDBG_E_IO; }
    if (bad.load()) return 0;
    const double t1 = timing ? now() : 0;
    const u64 total = link[C].base0, reads = link[C].read0;
    r->offs.resize(reads + 1);
    const std::function<void(int)> lay = [&](int i) {
        for (size_t c = (size_t)i; c < C; c += (size_t)T) { u64 o = link[c].base0; size_t j = link[c].read0; for (const mdbg_reader::FastRec& q : r->fast_recs[c]) { r->offs[j++] = o; o += q.len; } }
    };
    if (reads > 200000) { if (!r->pool->run(lay)) return MDBG_E_NOMEM; } else for (int i = 0; i < T; ++i) lay(i);
    r->offs[reads] = total;
    if (!ascii_out) {
        for (int i = 0; i < T; ++i) for (const Partial& q : parts[(size_t)i]) r->pw[q.word] = 0;       // words shared by records: clear, then OR the contributions
        for (int i = 0; i < T; ++i) for (const Partial& q : parts[(size_t)i]) r->pw[q.word] |= (u64)q.lo | ((u64)q.hi << 32);
        // the exception list in ascending position order: a worker's chunks are in file order, the workers' lists interleave
        r->pexc_pos.clear(); r->pexc_val.clear();
        size_t n_exc = 0; int lists = 0;
        for (int i = 0; i < T; ++i) { n_exc += ex[(size_t)i].pos.size(); lists += ex[(size_t)i].pos.empty() ? 0 : 1; }
        if (lists == 1) { for (int i = 0; i < T; ++i) if (!ex[(size_t)i].pos.empty()) { r->pexc_pos.swap(ex[(size_t)i].pos); r->pexc_val.swap(ex[(size_t)i].val); } }
        else if (lists > 1) {
            std::vector<std::pair<u64, u8>> all; all.reserve(n_exc);
            for (int i = 0; i < T; ++i) for (size_t e = 0; e < ex[(size_t)i].pos.size(); ++e) all.emplace_back(ex[(size_t)i].pos[e], ex[(size_t)i].val[e]);
            std::sort(all.begin(), all.end());
            r->pexc_pos.resize(n_exc); r->pexc_val.resize(n_exc);
            for (size_t e = 0; e < n_exc; ++e) { r->pexc_pos[e] = all[e].first; r->pexc_val[e] = all[e].second; }
        }
        r->packed_done = true;
    }
    if (timing) {
        double wsum = 0; for (double w : waited) wsum += w;
        fprintf(stderr, "[mdbg reader] window %.1f MB, %d threads, %zu chunks: scan + %s %.2f ms (waiting for the chunk in front: %.2f ms per thread), offsets + merge %.2f\n", (double)(w1 - w0) / 1e6, T, C,
                ascii_out ? "copy" : "pack", t1 - t0, wsum / T, now() - t1);
    }
    return 1;
}

extern "C" {
int mdbg_reader_next_packed(mdbg_reader* r, uint64_t max_bases, mdbg_packed_batch* out) {
    if (!r || !out) return MDBG_E_PARAM;
    memset(out, 0, sizeof *out);
    std::swap(r->pw, r->pw2); std::swap(r->pw_cap, r->pw2_cap); r->pexc_pos.swap(r->pexc_pos2); r->pexc_val.swap(r->pexc_val2);
    if (r->map) {
        int e;
        r->offs.swap(r->offs2);
        do e = reader_next_window(r, max_bases, false); while (!e && r->offs.size() == 1 && reader_more(r));
        if (e) return e;
        if (r->offs.size() == 1) { r->pexc_pos.clear(); r->pexc_val.clear(); out->offsets = r->offs.data(); return MDBG_OK; }      // end of file (the pieces still hold the last batch)
        const u64 total = r->offs.back();
        if (!r->packed_done) { e = reader_pack_pieces(r, total); if (e) return e; }
    } else {                                                   // streaming reader (.gz, .lz4, one thread): parse, then pack the ASCII batch
        const uint8_t* b; const uint64_t* o; uint64_t n;
        r->offs.swap(r->offs2);                                // the offsets handed out last stay intact during this call
        int e = mdbg_reader_next(r, max_bases, &b, &o, &n); if (e) return e;
        const u64 total = r->offs.back(), nw = (total + 31) / 32;
        if (nw + 8 > r->pw_cap && !r->grow_pw(nw + nw / 8 + 64)) return MDBG_E_NOMEM;
        ExcList ex;
        pack_range(b, total, 0, nw, r->pw, ex, __builtin_cpu_supports("avx2"));
        r->pexc_pos.swap(ex.pos); r->pexc_val.swap(ex.val);
    }
    out->words = r->pw; out->offsets = r->offs.data(); out->n_reads = r->offs.size() - 1;
    out->exc_pos = r->pexc_pos.data(); out->exc_val = r->pexc_val.data(); out->n_exc = r->pexc_pos.size();
    return MDBG_OK;
}

uint64_t mdbg_packed_words(uint64_t n_bases) { return (n_bases + 31) / 32; }

int mdbg_pack_reads(const uint8_t* bases, uint64_t n_bases, uint64_t* words, uint64_t* exc_pos, uint8_t* exc_val, uint64_t exc_cap,
                    uint64_t* n_exc, int threads) {
    if (!n_exc || (n_bases && (!bases || !words))) return MDBG_E_PARAM;
    const uint64_t nw = (n_bases + 31) / 32;
    const bool avx2 = __builtin_cpu_supports("avx2");
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > nw / 4096 + 1) threads = (int)(nw / 4096 + 1);
    std::vector<ExcList> ex((size_t)threads);
    if (threads == 1) pack_range(bases, n_bases, 0, nw, words, ex[0], avx2);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] { pack_range(bases, n_bases, nw * (uint64_t)t / threads, nw * (uint64_t)(t + 1) / threads, words, ex[(size_t)t], avx2); });
        for (auto& x : th) x.join();
    }
    uint64_t n = 0;
    for (auto& e : ex) for (size_t i = 0; i < e.pos.size(); ++i, ++n) if (n < exc_cap && exc_pos && exc_val) { exc_pos[n] = e.pos[i]; exc_val[n] = e.val[i]; }
    *n_exc = n;
    return n > exc_cap ? MDBG_E_CAPACITY : MDBG_OK;
}
// ---- --lmer-counts: the selected l-mers of a counts file ---------------------------------------------------------------------------
// What main.rs:544-566 (the table: key = min(lmer, revcomp(lmer)), later lines win) and minimizers::minimizers_preparation
// (minimizers.rs:53-113, the branch with counts) do, ending in the list of 2-bit codes mdbg_set_lmer_filter takes (both orientations).
namespace {
const u64 NTH_A = 0x3c8bfbb395c60474ull, NTH_C = 0x3193c18562a02b4cull, NTH_G = 0x20323ed082572324ull, NTH_T = 0x295549f54be24456ull;
inline u64 rol64h(u64 x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
inline u64 nth_seed(char c) { return c == 'A' ? NTH_A : c == 'C' ? NTH_C : c == 'G' ? NTH_G : c == 'T' ? NTH_T : 0; }
// canonical ntHash of s[0..l) over ACGT: min(XOR_i rol(h(s_i), l-1-i), XOR_i rol(h(comp s_i), i))
u64 ntc64_acgt(const std::string& s, u32 l) {
    u64 f = 0, r = 0;
    for (u32 i = 0; i < l; ++i) {
        const char c = s[i], cc = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        f ^= rol64h(nth_seed(c), l - 1 - i); r ^= rol64h(nth_seed(cc), i);
    }
    return f < r ? f : r;
}
std::string revcomp_str(const std::string& s) { std::string o(s.rbegin(), s.rend()); for (char& c : o) c = switch_base(c); return o; }
}  // namespace

int mdbg_lmer_filter_from_counts(const char* path, uint32_t l, double density, uint32_t count_min, uint32_t count_max,
                                 uint64_t** codes, uint64_t* n_codes, uint64_t* n_ignored) {
    if (!path || !codes || !n_codes || l < 1 || l > 32) return MDBG_E_PARAM;
    *codes = nullptr; *n_codes = 0; if (n_ignored) *n_ignored = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return MDBG_E_IO;
    std::unordered_map<std::string, u32> counts;                           // main.rs:544
    u64 ignored = 0;
    std::string line; char buf[1 << 16]; bool bad = false;
    auto take = [&](const std::string& ln) {
        size_t a = 0; while (a < ln.size() && isspace((unsigned char)ln[a])) ++a;
        if (a == ln.size()) { if (!ln.empty()) bad = true; return; }       // a blank line: vec[0] panics in the reference
        size_t b = a; while (b < ln.size() && !isspace((unsigned char)ln[b])) ++b;
        size_t c0 = b; while (c0 < ln.size() && isspace((unsigned char)ln[c0])) ++c0;
        size_t c1 = c0; while (c1 < ln.size() && !isspace((unsigned char)ln[c1])) ++c1;
        if (c0 == c1) { bad = true; return; }
        char* end = nullptr; errno = 0;
        const std::string num = ln.substr(c0, c1 - c0);
        const unsigned long long v = strtoull(num.c_str(), &end, 10);
        if (errno || *end || num[0] == '-' || v > 0xFFFFFFFFull) { bad = true; return; }       // parse::<u32>().unwrap()
        std::string lmer = ln.substr(a, b - a);
        const std::string rc = revcomp_str(lmer);
        counts[lmer < rc ? lmer : rc] = (u32)v;                            // main.rs:560-564
    };
    while (fgets(buf, sizeof buf, f)) {
        line += buf;
        if (!line.empty() && line.back() == '\n') { take(line); line.clear(); }
    }
    if (!line.empty()) take(line);
    const bool rerr = ferror(f) != 0;
    fclose(f);
    if (rerr) return MDBG_E_IO;
    if (bad) return MDBG_E_PARAM;
    std::vector<u64> out;
    for (const auto& kv : counts) {
        const std::string& lmer = kv.first;
        bool acgt = lmer.size() == l;                                      // another length can never equal a read's l-mer (read.rs:198, 202)
        for (size_t i = 0; acgt && i < lmer.size(); ++i) acgt = lmer[i] == 'A' || lmer[i] == 'C' || lmer[i] == 'G' || lmer[i] == 'T';
        if (!acgt) { ++ignored; continue; }                                // not representable as a 2-bit code; no k-mer counter lists such l-mers
        const bool skip = kv.second >= count_max || kv.second <= count_min;          // minimizers.rs:80
        double hn = (double)ntc64_acgt(lmer, l) / 18446744073709551616.0;             // :88-90 (u64::MAX as f64 == 2^64)
        if (skip) hn = 1.0;                                                // :91-95
        if (!(hn <= density)) continue;                                    // :96
        const std::string rc = revcomp_str(lmer);
        for (const std::string* w : {&lmer, &rc}) {                        // :99-105: the l-mer and its reverse complement
            u64 code = 0;
            for (char ch : *w) code = (code << 2) | (u64)(((unsigned char)ch >> 1) & 3);
            out.push_back(code);
        }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());             // a palindromic l-mer is its own reverse complement
    u64* mem = (u64*)malloc((out.size() + 1) * sizeof(u64));
    if (!mem) return MDBG_E_NOMEM;
    memcpy(mem, out.data(), out.size() * sizeof(u64));
    *codes = mem; *n_codes = out.size(); if (n_ignored) *n_ignored = ignored;
    return MDBG_OK;
}
void mdbg_lmer_filter_free(uint64_t* codes) { free(codes); }

}  // extern "C"
