// the gzip decoder alone (gz::GzIn::read into a fixed buffer, one thread): g++ -O2 -std=c++17 -pthread -o scratch/ubench/gz_prof scratch/ubench/gz_prof.cpp -lz; gz_prof <file.gz> [reps]
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "../../rust_mdbg_amd/csrc/gz_inflate.h"
static std::vector<uint8_t> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<uint8_t> v; uint8_t b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    std::vector<uint8_t> c = slurp(argv[1]);
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<uint8_t> buf(1 << 20);
    double best = 1e9; size_t total = 0;
    for (int r = 0; r < reps; ++r) {
        gz::GzIn g; g.open(c.data(), c.size(), 1);
        double t = now(); total = 0;
        for (;;) { const int n = g.read(buf.data(), buf.size()); if (n <= 0) break; total += n; }
        best = std::min(best, now() - t);
    }
    printf("%s: %zu bytes, best %.1f ms = %.0f MB/s\n", argv[1], total, best * 1e3, total / best / 1e6);
}
