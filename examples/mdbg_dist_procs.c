/* mdbg_dist_procs.c — the multi-GPU layer (include/mdbg_dist.h) driven by W separate OS PROCESSES, one mdbg_dist each, sharing the
 * one GPU of a test box.  The communicator is a function table implemented right here over a shared-memory segment: host-staged
 * `exchange` (sender copies device -> segment, process-shared barrier, receiver copies segment -> device), `allgather_u64` and
 * `allreduce_sum_u64` through the same segment.  On a real node every rank is a process with its own GPU and the table comes from
 * mdbg_comm_rccl(ncclComm_t); what this program adds over mdbg_dist_threads.c is everything threads hide: per-process library state,
 * handle lifetimes, ordering of the collectives across address spaces.
 *
 * The parent maps the segment, forks the ranks BEFORE anything touches the GPU runtime, waits for them, and only then builds the
 * reference: ONE context fed all reads with the same ordinals.  The partitions (left in the segment by the ranks) are put together
 * by their global row and compared with it field by field.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/mdbg_dist_procs.c -o mdbg_dist_procs -Lrust_mdbg_amd -lmdbg_hip -lpthread
 *   ./mdbg_dist_procs [world=2] [reads_per_rank=300] [rounds=2] [packed=0] [chunks=1]
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "mdbg_dist.h"

#define MAXW 8
#define MAXX 64                               /* transfers one rank lists per exchange */
#define STAGE_BYTES ((uint64_t)48 << 20)      /* staging area per rank */
#define PART_BYTES ((uint64_t)48 << 20)       /* result area per rank */
#define AG_MAX 4096
#define CHECK(x) do { int e_ = (x); if (e_) { fprintf(stderr, "[rank %d] %s:%d: %s -> %d (%s)\n", g_rank, __FILE__, __LINE__, #x, e_, mdbg_strerror(e_)); _exit(2); } } while (0)
static int g_rank = -1;

/* ---- the shared segment ----------------------------------------------------------------------------------------------------- */
typedef struct xdesc_t { uint32_t peer; uint64_t bytes, off; } xdesc_t;        /* one staged send: destination, size, offset in the sender's staging area */
typedef struct seg_t {
    pthread_barrier_t bar;
    uint32_t world;
    uint64_t ag[MAXW][AG_MAX];
    uint32_t n_sends[MAXW]; xdesc_t sends[MAXW][MAXX];
    uint64_t part_n[MAXW], part_global[MAXW], part_distinct[MAXW], part_k[MAXW];
    int failed[MAXW];
} seg_t;
static unsigned char* stage_of(seg_t* s, uint32_t r) { return (unsigned char*)s + ((sizeof(seg_t) + 4095) & ~(size_t)4095) + (size_t)r * STAGE_BYTES; }
static unsigned char* part_of(seg_t* s, uint32_t r) { return stage_of(s, s->world) + (size_t)r * PART_BYTES; }
static size_t seg_bytes(uint32_t w) { return ((sizeof(seg_t) + 4095) & ~(size_t)4095) + (size_t)w * (STAGE_BYTES + PART_BYTES); }

typedef struct rank_t { seg_t* s; uint32_t rank; mdbg_ctx* ctx; } rank_t;
static void bar(seg_t* s) { pthread_barrier_wait(&s->bar); }

static int p_allgather(void* self, const uint64_t* send, uint32_t n, uint64_t* recv) {
    rank_t* r = (rank_t*)self; seg_t* s = r->s;
    if (n > AG_MAX) return MDBG_E_PARAM;
    memcpy(s->ag[r->rank], send, (size_t)n * 8);
    bar(s);
    for (uint32_t p = 0; p < s->world; ++p) memcpy(recv + (size_t)p * n, s->ag[p], (size_t)n * 8);
    bar(s);
    return 0;
}
static int p_exchange(void* self, const mdbg_xfer* sends, uint32_t ns, const mdbg_xfer* recvs, uint32_t nr) {
    rank_t* r = (rank_t*)self; seg_t* s = r->s;
    if (ns > MAXX) return MDBG_E_PARAM;
    uint64_t off = 0;
    for (uint32_t i = 0; i < ns; ++i) {                      /* my sends: device -> my staging area */
        if (off + sends[i].bytes > STAGE_BYTES) { fprintf(stderr, "[rank %u] exchange: staging area too small\n", r->rank); return MDBG_E_NOMEM; }
        CHECK(mdbg_copy_to_host(r->ctx, stage_of(s, r->rank) + off, sends[i].d_ptr, sends[i].bytes));
        s->sends[r->rank][i].peer = sends[i].peer; s->sends[r->rank][i].bytes = sends[i].bytes; s->sends[r->rank][i].off = off;
        off += (sends[i].bytes + 63) & ~(uint64_t)63;
    }
    s->n_sends[r->rank] = ns;
    bar(s);
    uint32_t cursor[MAXW]; memset(cursor, 0, sizeof cursor);
    int rc = 0;
    for (uint32_t i = 0; i < nr && !rc; ++i) {               /* the i-th receive from peer p pairs with p's next send to me */
        const uint32_t p = recvs[i].peer;
        uint32_t j = cursor[p];
        while (j < s->n_sends[p] && s->sends[p][j].peer != r->rank) ++j;
        if (j == s->n_sends[p] || s->sends[p][j].bytes != recvs[i].bytes) { fprintf(stderr, "[rank %u] exchange: unmatched transfer\n", r->rank); rc = MDBG_E_STATE; break; }
        cursor[p] = j + 1;
        rc = mdbg_copy_to_device(r->ctx, recvs[i].d_ptr, stage_of(s, p) + s->sends[p][j].off, recvs[i].bytes);
    }
    bar(s);                                                  /* senders may reuse their staging area only now */
    return rc;
}
static int p_allreduce(void* self, uint64_t* d_buf, uint64_t n) {
    rank_t* r = (rank_t*)self; seg_t* s = r->s;
    if (n * 8 > STAGE_BYTES) return MDBG_E_NOMEM;
    uint64_t* mine = (uint64_t*)stage_of(s, r->rank);
    if (n) CHECK(mdbg_copy_to_host(r->ctx, mine, d_buf, n * 8));
    bar(s);
    uint64_t* acc = (uint64_t*)calloc(n ? n : 1, 8);
    for (uint32_t p = 0; p < s->world; ++p) { const uint64_t* q = (const uint64_t*)stage_of(s, p); for (uint64_t i = 0; i < n; ++i) acc[i] += q[i]; }
    bar(s);                                                  /* everybody has read every buffer */
    int rc = n ? mdbg_copy_to_device(r->ctx, d_buf, acc, n * 8) : 0;
    free(acc);
    return rc;
}

/* ---- one rank = one process ------------------------------------------------------------------------------------------------ */
typedef struct job_t { mdbg_params P; uint64_t reads_per_rank, genome; int rounds, packed, chunks; } job_t;
static void synth(mdbg_synth_params* sp, uint64_t genome, uint64_t n) {
    memset(sp, 0, sizeof *sp);
    sp->seed = 7; sp->genome_len = genome; sp->n_reads = n; sp->mean_len = 9000; sp->sd_len = 1500; sp->min_len = 2000; sp->max_len = 16000; sp->err_ppm = 2000;
}
static void fetch(mdbg_ctx* c, void* dst, const void* src, uint64_t bytes) { if (bytes) CHECK(mdbg_copy_to_host(c, dst, src, bytes)); }

static int rank_main(seg_t* s, uint32_t rank, const job_t* j) {
    g_rank = (int)rank;
    rank_t rk; rk.s = s; rk.rank = rank; rk.ctx = NULL;
    mdbg_comm comm; memset(&comm, 0, sizeof comm);          /* exchange_begin / exchange_wait stay NULL: this transport only has the blocking form */
    comm.self = &rk; comm.rank = rank; comm.world = s->world;
    comm.allgather_u64 = p_allgather; comm.exchange = p_exchange; comm.allreduce_sum_u64 = p_allreduce;
    int err = 0;
    mdbg_dist* d = mdbg_dist_create(&j->P, &comm, &err);
    if (!d) { fprintf(stderr, "[rank %u] mdbg_dist_create: %d (%s)\n", rank, err, mdbg_strerror(err)); return 2; }
    if (j->chunks > 1) CHECK(mdbg_dist_set_pipeline(d, (uint32_t)j->chunks));
    mdbg_ctx* c = mdbg_dist_ctx(d);
    rk.ctx = c;
    mdbg_ctx* gen = mdbg_create(&j->P, &err);               /* generates this rank's reads (its buffers must outlive the rounds) */
    if (!gen) return 2;
    const uint64_t per_round = j->reads_per_rank / (uint64_t)j->rounds;
    for (int rd = 0; rd < j->rounds; ++rd) {
        /* the last rank sits out the last round: every rank still takes part in the collective */
        const int idle = (rd == j->rounds - 1 && rank == s->world - 1 && s->world > 1);
        const uint64_t first = ((uint64_t)rd * s->world + rank) * per_round;       /* global ordinal of the batch's first read */
        mdbg_synth_params sp; synth(&sp, j->genome, per_round);
        const uint8_t* db = NULL; const uint64_t* dof = NULL; uint64_t nb = 0;
        CHECK(mdbg_synth_reads_device(gen, &sp, first, &db, &dof, &nb));
        if (idle) { CHECK(mdbg_dist_ingest_batch_device(d, NULL, NULL, 0, 0, 0)); continue; }
        if (j->packed) {
            uint8_t* hb = (uint8_t*)malloc(nb);
            fetch(gen, hb, db, nb);
            const uint64_t nw = (nb + 31) / 32;
            uint64_t* words = (uint64_t*)calloc(nw ? nw : 1, 8);
            for (uint64_t q = 0; q < nb; ++q) { words[q >> 5] |= (uint64_t)((hb[q] >> 1) & 1u) << (q & 31); words[q >> 5] |= (uint64_t)((hb[q] >> 2) & 1u) << (32 + (q & 31)); }
            CHECK(mdbg_copy_to_device(gen, (void*)db, words, nw * 8));       /* the ASCII bases are no longer needed: their buffer takes the words */
            mdbg_packed_batch pb; memset(&pb, 0, sizeof pb);
            pb.words = (const uint64_t*)db; pb.offsets = dof; pb.n_reads = per_round;
            CHECK(mdbg_dist_ingest_batch_packed_device(d, &pb, nb, first));
            free(hb); free(words);
        } else CHECK(mdbg_dist_ingest_batch_device(d, db, dof, per_round, nb, first));
    }
    mdbg_nodes nd; const uint64_t* d_row = NULL; uint64_t ng = 0;
    CHECK(mdbg_dist_finalize(d, &nd, &d_row, &ng));
    /* partition -> segment: keys | src_read | row | src_start | src_end | shift_full | index | seqlen | abundance */
    const uint64_t need = nd.n * (nd.k * 8 + 4 + 2 + 4 + 8 + 8 + 8 + 8 + 16) + 64;
    if (need > PART_BYTES) { fprintf(stderr, "[rank %u] result area too small\n", rank); return 3; }
    unsigned char* o = part_of(s, rank);
    s->part_n[rank] = nd.n; s->part_global[rank] = ng; s->part_distinct[rank] = nd.n_distinct; s->part_k[rank] = nd.k;
    fetch(c, o, nd.keys, nd.n * nd.k * 8); o += nd.n * nd.k * 8;
    fetch(c, o, nd.src_read, nd.n * 8); o += nd.n * 8;
    fetch(c, o, d_row, nd.n * 8); o += nd.n * 8;
    fetch(c, o, nd.src_start, nd.n * 8); o += nd.n * 8;
    fetch(c, o, nd.src_end, nd.n * 8); o += nd.n * 8;
    fetch(c, o, nd.shift_full, nd.n * 16); o += nd.n * 16;
    fetch(c, o, nd.index, nd.n * 4); o += nd.n * 4;
    fetch(c, o, nd.seqlen, nd.n * 4); o += nd.n * 4;
    fetch(c, o, nd.abundance, nd.n * 2);
    bar(s);
    mdbg_destroy(gen);
    mdbg_dist_destroy(d);
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 2;
    const uint64_t rpr = argc > 2 ? strtoull(argv[2], NULL, 10) : 300;
    const int rounds = argc > 3 ? atoi(argv[3]) : 2;
    const int packed = argc > 4 ? atoi(argv[4]) : 0;
    const int chunks = argc > 5 ? atoi(argv[5]) : 1;
    if (W < 1 || W > MAXW || rounds < 1 || rpr % (uint64_t)rounds || chunks < 1 || chunks > 64) { fprintf(stderr, "bad arguments\n"); return 1; }
    job_t job; memset(&job, 0, sizeof job);
    job.P.k = 9; job.P.l = 12; job.P.density = 0.004; job.P.min_abundance = 2; job.P.device = -1;
    job.reads_per_rank = rpr; job.genome = 150000; job.rounds = rounds; job.packed = packed; job.chunks = chunks;

    seg_t* s = (seg_t*)mmap(NULL, seg_bytes(W), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (s == MAP_FAILED) { perror("mmap"); return 1; }
    s->world = W;
    pthread_barrierattr_t ba; pthread_barrierattr_init(&ba); pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&s->bar, &ba, W);
    pid_t pid[MAXW];
    for (uint32_t r = 0; r < W; ++r) {                      /* the GPU runtime has not been touched yet: every child initialises its own */
        pid[r] = fork();
        if (pid[r] < 0) { perror("fork"); return 1; }
        if (pid[r] == 0) _exit(rank_main(s, r, &job));
    }
    int bad = 0;
    for (uint32_t r = 0; r < W; ++r) { int st = 0; waitpid(pid[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { fprintf(stderr, "rank %u (pid %d) failed: status 0x%x\n", r, (int)pid[r], st); bad = 1; } }
    if (bad) return 5;

    /* reference: ONE context over the same reads (same ordinals), in this process */
    int err = 0;
    mdbg_ctx* one = mdbg_create(&job.P, &err), *gen = mdbg_create(&job.P, &err);
    if (!one || !gen) { fprintf(stderr, "mdbg_create: %d\n", err); return 2; }
    const uint64_t per_round = rpr / (uint64_t)rounds;
    for (int rd = 0; rd < rounds; ++rd) for (uint32_t r = 0; r < W; ++r) {
        if (rd == rounds - 1 && r == W - 1 && W > 1) continue;
        const uint64_t first = ((uint64_t)rd * W + r) * per_round;
        mdbg_synth_params sp; synth(&sp, job.genome, per_round);
        const uint8_t* db; const uint64_t* dof; uint64_t nb;
        CHECK(mdbg_synth_reads_device(gen, &sp, first, &db, &dof, &nb));
        CHECK(mdbg_ingest_batch_device(one, db, dof, per_round, nb, first));
    }
    mdbg_nodes ref;
    CHECK(mdbg_finalize(one, &ref));
    uint64_t total = 0;
    for (uint32_t r = 0; r < W; ++r) total += s->part_n[r];
    int ok = total == ref.n && ref.n > 100;
    unsigned char* seen = (unsigned char*)calloc(ref.n + 1, 1);
    for (uint32_t r = 0; r < W && ok; ++r) {
        const uint64_t n = s->part_n[r];
        ok = ok && s->part_global[r] == ref.n && s->part_distinct[r] == ref.n_distinct && s->part_k[r] == ref.k;
        const unsigned char* o = part_of(s, r);
        const uint64_t* keys = (const uint64_t*)o; o += n * ref.k * 8;
        const uint64_t* src_read = (const uint64_t*)o; o += n * 8;
        const uint64_t* row = (const uint64_t*)o; o += n * 8;
        const uint64_t* src_start = (const uint64_t*)o; o += n * 8;
        const uint64_t* src_end = (const uint64_t*)o; o += n * 8;
        const uint64_t* shift_full = (const uint64_t*)o; o += n * 16;
        const uint32_t* index = (const uint32_t*)o; o += n * 4;
        const uint32_t* seqlen = (const uint32_t*)o; o += n * 4;
        const uint16_t* abundance = (const uint16_t*)o;
        for (uint64_t i = 0; i < n && ok; ++i) {
            const uint64_t rw = row[i];
            ok = rw < ref.n && !seen[rw] && index[i] == ref.index[rw] && abundance[i] == ref.abundance[rw] && seqlen[i] == ref.seqlen[rw] &&
                 src_read[i] == ref.src_read[rw] && !memcmp(keys + i * ref.k, ref.keys + rw * ref.k, ref.k * 8) &&
                 /* what needs the raw positions of the A-th sighting — fetched from the rank that sketched the read */
                 src_start[i] == ref.src_start[rw] && src_end[i] == ref.src_end[rw] && !memcmp(shift_full + 2 * i, ref.shift_full + 2 * rw, 16);
            if (rw < ref.n) seen[rw] = 1;
        }
    }
    printf("world %u PROCESSES, %llu reads per rank in %d rounds (x %d pipelined chunks), %s input: %llu nodes (%llu distinct k-min-mers); partitions", W,
           (unsigned long long)rpr, rounds, chunks, packed ? "packed" : "ASCII", (unsigned long long)ref.n, (unsigned long long)ref.n_distinct);
    for (uint32_t r = 0; r < W; ++r) printf(" %llu", (unsigned long long)s->part_n[r]);
    printf(" -> %s\n", ok ? "EQUAL to the single-context table" : "MISMATCH");
    mdbg_destroy(one); mdbg_destroy(gen);
    return ok ? 0 : 4;
}
