// scratch/gz_test.cpp — gz_inflate.h against zlib: equality of the inflated bytes, speed, and behaviour on damaged streams.
//   g++ -O2 -std=c++17 -pthread -o /tmp/gzt/gz_test scratch/gz_test.cpp -lz
//   gz_test cmp <file.gz> [threads]      inflate with both, compare, print MB/s
//   gz_test fuzz <file.gz> <n> <seed>    n damaged copies (bit flips, truncations): must end with an error or with zlib's bytes, never crash
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <random>
#include <zlib.h>
#include "../rust_mdbg_amd/csrc/gz_inflate.h"
static std::vector<uint8_t> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<uint8_t> v; uint8_t b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// zlib, all members; ok = false on error
static std::vector<uint8_t> zl(const std::vector<uint8_t>& c, bool& ok) {
    std::vector<uint8_t> out; ok = true;
    size_t at = 0;
    while (at < c.size()) {
        z_stream s; memset(&s, 0, sizeof s);
        if (inflateInit2(&s, 31) != Z_OK) { ok = false; break; }
        s.next_in = (Bytef*)c.data() + at; s.avail_in = (uInt)std::min<size_t>(c.size() - at, 1u << 30);
        int r;
        do {
            uint8_t buf[1 << 16]; s.next_out = buf; s.avail_out = sizeof buf;
            r = inflate(&s, Z_NO_FLUSH);
            out.insert(out.end(), buf, buf + (sizeof buf - s.avail_out));
        } while (r == Z_OK);
        const size_t used = (c.size() - at < (1u << 30) ? c.size() - at : (1u << 30)) - s.avail_in;
        inflateEnd(&s);
        if (r != Z_STREAM_END) { ok = false; break; }
        at += used;
        if (at + 2 <= c.size() && !(c[at] == 0x1f && c[at + 1] == 0x8b)) break;      // trailing garbage: ignored
    }
    return out;
}
static std::vector<uint8_t> mine(const std::vector<uint8_t>& c, int threads, bool& ok, size_t chunk = 1 << 20) {
    gz::GzIn g; g.open(c.data(), c.size(), threads);
    std::vector<uint8_t> out, buf(chunk);
    for (;;) { const int n = g.read(buf.data(), buf.size()); if (n <= 0) { ok = n == 0; break; } out.insert(out.end(), buf.begin(), buf.begin() + n); }
    g.close_pool();
    return out;
}
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::vector<uint8_t> c = slurp(argv[2]);
    if (!strcmp(argv[1], "cmp")) {
        const int threads = argc > 3 ? atoi(argv[3]) : 1;
        bool ok1, ok2;
        double t = now(); std::vector<uint8_t> a = zl(c, ok1); const double tz = now() - t;
        t = now(); std::vector<uint8_t> b = mine(c, threads, ok2); const double tm = now() - t;
        printf("%s: zlib %s %zu bytes %.0f MB/s | gz_inflate(%d) %s %zu bytes %.0f MB/s | %s\n", argv[2], ok1 ? "ok" : "ERR", a.size(), a.size() / tz / 1e6, threads, ok2 ? "ok" : "ERR",
               b.size(), b.size() / tm / 1e6, ok1 == ok2 && a == b ? "EQUAL" : "DIFFERENT");
        // odd read sizes exercise the chunk boundaries
        bool ok3; std::vector<uint8_t> d = mine(c, threads, ok3, 7919);
        if (ok3 != ok2 || d != b) { printf("read-size dependence!\n"); return 1; }
        return ok1 == ok2 && a == b ? 0 : 1;
    }
    if (!strcmp(argv[1], "fuzz")) {
        const int n = atoi(argv[3]); std::mt19937_64 rng(argc > 4 ? atoll(argv[4]) : 1);
        int errs = 0, same = 0, differ = 0;
        for (int i = 0; i < n; ++i) {
            std::vector<uint8_t> d = c;
            const int kind = rng() % 4;
            if (kind == 0) d.resize(rng() % d.size());
            else { const int flips = 1 + rng() % 4; for (int f = 0; f < flips; ++f) { const size_t p = rng() % d.size(); d[p] ^= (uint8_t)(1u << (rng() % 8)); } }
            if (kind == 3) { const size_t p = rng() % d.size(), l = std::min<size_t>(d.size() - p, 1 + rng() % 64); for (size_t j = 0; j < l; ++j) d[p + j] = (uint8_t)rng(); }
            // exact-size heap copy: reads past the end are visible to a sanitizer build
            uint8_t* h = (uint8_t*)malloc(d.size() ? d.size() : 1); memcpy(h, d.data(), d.size());
            gz::GzIn g; g.open(h, d.size(), 1 + (int)(rng() % 6));
            std::vector<uint8_t> out, buf(65536); bool ok = true;
            for (;;) { const int r = g.read(buf.data(), buf.size()); if (r <= 0) { ok = r == 0; break; } out.insert(out.end(), buf.begin(), buf.begin() + r); if (out.size() > (c.size() + 1000) * 1100) { ok = false; break; } }
            g.close_pool(); free(h);
            if (!ok) { ++errs; continue; }
            bool okz; std::vector<uint8_t> z = zl(d, okz);
            if (okz && z == out) ++same; else { ++differ; if (okz) printf("case %d: both accept, bytes differ (%zu vs %zu)\n", i, out.size(), z.size()); }
        }
        printf("fuzz: %d cases: %d rejected, %d accepted with zlib's bytes, %d accepted where zlib differs or rejects\n", n, errs, same, differ);
        return 0;
    }
    return 2;
}
