// scratch/reader_check.cpp — the host reader under a sanitizer: reads a file with T threads (both batch calls), prints record count and a checksum.
//   g++ -O1 -g -fsanitize=thread -std=c++17 -pthread -o /tmp/gzt/reader_tsan scratch/reader_check.cpp rust_mdbg_amd/csrc/mdbg_emit.cpp -lz
//   reader_tsan <file> <threads> [max_bases]
#include <cstdio>
#include <cstdlib>
#include "../include/mdbg_emit.h"
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int T = atoi(argv[2]); const uint64_t mb = argc > 3 ? strtoull(argv[3], nullptr, 10) : (64ull << 20);
    for (int packed = 0; packed < 2; ++packed) {
        int err = 0;
        mdbg_reader* r = mdbg_reader_open_mt(argv[1], 0, T, &err);
        if (!r) { printf("open failed %d\n", err); return 1; }
        uint64_t nrec = 0, nb = 0, sum = 0;
        for (;;) {
            if (!packed) {
                const uint8_t* b; const uint64_t* o; uint64_t n;
                const int e = mdbg_reader_next(r, mb, &b, &o, &n); if (e) { printf("error %d\n", e); return 1; }
                if (!n) break;
                nrec += n; nb += o[n];
                for (uint64_t i = 0; i < o[n]; i += 97) sum = sum * 31 + b[i];
            } else {
                mdbg_packed_batch pb;
                const int e = mdbg_reader_next_packed(r, mb, &pb); if (e) { printf("error %d\n", e); return 1; }
                if (!pb.n_reads) break;
                nrec += pb.n_reads; nb += pb.offsets[pb.n_reads];
                for (uint64_t i = 0; i < (pb.offsets[pb.n_reads] + 31) / 32; i += 13) sum = sum * 31 + pb.words[i];
            }
        }
        mdbg_reader_close(r);
        printf("%s T=%d %s: %llu records, %llu bases, checksum %016llx\n", argv[1], T, packed ? "packed" : "ascii", (unsigned long long)nrec, (unsigned long long)nb, (unsigned long long)sum);
    }
}
