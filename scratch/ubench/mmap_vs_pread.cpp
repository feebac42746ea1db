// how should many threads read a file from the page cache ONCE: through a mapping (a minor fault per 16 pages, page-table work, munmap later) or with pread into a
// small per-thread buffer that stays in L2 (a kernel copy per byte, no faults)?   usage: mmap_vs_pread FILE THREADS...
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static size_t count_nl(const unsigned char* p, size_t n) { size_t c = 0; const unsigned char* e = p + n; while (p < e) { const void* q = memchr(p, '\n', (size_t)(e - p)); if (!q) break; ++c; p = (const unsigned char*)q + 1; } return c; }
int main(int argc, char** argv) {
    const char* path = argv[1];
    const int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); const size_t n = (size_t)st.st_size;
    const size_t S = 256u << 10;
    for (int a = 2; a < argc; ++a) {
        const int T = atoi(argv[a]);
        for (int mode = 0; mode < 4; ++mode) {      // 0 mmap, 1 pread, 2 mmap + MADV_POPULATE_READ per 64 MB on one helper thread ahead, 3 mmap with 2-MB-aligned hugepage advice
            if (mode == 3) continue;
            const double t0 = now();
            unsigned char* m = nullptr;
            if (mode != 1) m = (unsigned char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            std::atomic<size_t> next{0}; std::atomic<size_t> lines{0};
            std::thread helper;
            if (mode == 2) helper = std::thread([&] { for (size_t o = 0; o < n; o += 64u << 20) madvise(m + o, std::min<size_t>(64u << 20, n - o), 22 /* MADV_POPULATE_READ */); });
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&] {
                std::vector<unsigned char> buf(mode == 1 ? S : 0);
                size_t c = 0;
                for (;;) {
                    const size_t o = next.fetch_add(S); if (o >= n) break;
                    const size_t len = std::min(S, n - o);
                    if (mode == 1) { size_t got = 0; while (got < len) { const ssize_t r = pread(fd, buf.data() + got, len - got, (off_t)(o + got)); if (r <= 0) break; got += (size_t)r; } c += count_nl(buf.data(), len); }
                    else c += count_nl(m + o, len);
                }
                lines += c;
            });
            for (auto& x : th) x.join();
            const double t1 = now();
            if (helper.joinable()) helper.join();
            if (m) munmap(m, n);
            const double t2 = now();
            printf("%2d threads, %-28s: scan %.3f s (%.1f GB/s), + unmap %.3f s, %zu lines\n", T, mode == 0 ? "mmap" : mode == 1 ? "pread into 256-KB buffers" : "mmap + populate helper", t1 - t0, n / (t1 - t0) / 1e9, t2 - t1, lines.load());
        }
    }
    return 0;
}
