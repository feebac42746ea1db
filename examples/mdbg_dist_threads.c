/* mdbg_dist_threads.c — the multi-GPU layer (include/mdbg_dist.h) driven from plain C: W ranks as threads of one process, each with
 * its own mdbg_dist context, exchanging over a communicator implemented right here with a barrier and staged copies (on a real
 * node every rank is a process with its own GPU and the communicator is mdbg_comm_rccl(ncclComm_t)).  The partitions the ranks
 * return are put together by their global row and compared, field by field, with the table of ONE context fed all reads.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/mdbg_dist_threads.c -o mdbg_dist_threads -Lrust_mdbg_amd -lmdbg_hip -lpthread
 *   ./mdbg_dist_threads [world=2] [reads_per_rank=300] [rounds=2] [packed=0]
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdbg_dist.h"

#define MAXW 16
#define CHECK(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, e_, mdbg_strerror(e_)); exit(2); } } while (0)

/* ---- a communicator for ranks that are threads of one process ------------------------------------------------------------- */
typedef struct world_t {
    uint32_t world;
    pthread_barrier_t bar;
    const uint64_t* ag_send[MAXW]; uint32_t ag_n;
    const mdbg_xfer* sends[MAXW]; uint32_t n_sends[MAXW];
    uint64_t* ar_buf[MAXW];
    mdbg_ctx* ctx[MAXW];                /* for the staged copies */
} world_t;
typedef struct rank_t { world_t* w; uint32_t rank; } rank_t;

static int t_allgather(void* self, const uint64_t* send, uint32_t n, uint64_t* recv) {
    rank_t* r = (rank_t*)self; world_t* w = r->w;
    w->ag_send[r->rank] = send;
    pthread_barrier_wait(&w->bar);
    for (uint32_t p = 0; p < w->world; ++p) memcpy(recv + (size_t)p * n, w->ag_send[p], (size_t)n * 8);
    pthread_barrier_wait(&w->bar);
    return 0;
}
static int t_exchange(void* self, const mdbg_xfer* sends, uint32_t ns, const mdbg_xfer* recvs, uint32_t nr) {
    rank_t* r = (rank_t*)self; world_t* w = r->w;
    w->sends[r->rank] = sends; w->n_sends[r->rank] = ns;
    pthread_barrier_wait(&w->bar);
    uint32_t cursor[MAXW]; memset(cursor, 0, sizeof cursor);
    for (uint32_t i = 0; i < nr; ++i) {                      /* the i-th receive from peer p pairs with p's next send to me */
        const uint32_t p = recvs[i].peer;
        uint32_t j = cursor[p];
        while (j < w->n_sends[p] && w->sends[p][j].peer != r->rank) ++j;
        if (j == w->n_sends[p] || w->sends[p][j].bytes != recvs[i].bytes) { fprintf(stderr, "exchange: unmatched transfer\n"); exit(3); }
        cursor[p] = j + 1;
        void* stage = malloc(recvs[i].bytes);
        /* (through my own context: the peer holds its context's lock while it waits at the barrier) */
        CHECK(mdbg_copy_to_host(w->ctx[r->rank], stage, w->sends[p][j].d_ptr, recvs[i].bytes));
        CHECK(mdbg_copy_to_device(w->ctx[r->rank], recvs[i].d_ptr, stage, recvs[i].bytes));
        free(stage);
    }
    pthread_barrier_wait(&w->bar);                          /* senders may reuse their buffers only now */
    return 0;
}
static int t_allreduce(void* self, uint64_t* d_buf, uint64_t n) {
    rank_t* r = (rank_t*)self; world_t* w = r->w;
    w->ar_buf[r->rank] = d_buf;
    pthread_barrier_wait(&w->bar);
    uint64_t* acc = (uint64_t*)calloc(n ? n : 1, 8), *tmp = (uint64_t*)malloc((n ? n : 1) * 8);
    for (uint32_t p = 0; p < w->world; ++p) {
        if (n) CHECK(mdbg_copy_to_host(w->ctx[r->rank], tmp, w->ar_buf[p], n * 8));
        for (uint64_t i = 0; i < n; ++i) acc[i] += tmp[i];
    }
    pthread_barrier_wait(&w->bar);                          /* everybody has read every buffer */
    if (n) CHECK(mdbg_copy_to_device(w->ctx[r->rank], d_buf, acc, n * 8));
    free(acc); free(tmp);
    pthread_barrier_wait(&w->bar);
    return 0;
}

/* ---- one rank -------------------------------------------------------------------------------------------------------------- */
typedef struct part_t { uint64_t n, n_global, n_distinct; uint64_t* keys; uint32_t* index; uint16_t* abundance; uint32_t* seqlen; uint64_t* src_read; uint64_t* row;
                        uint64_t* src_start; uint64_t* src_end; uint64_t* shift_full; } part_t;
typedef struct job_t { rank_t rk; mdbg_params P; uint64_t reads_per_rank, genome; int rounds, packed, chunks, whole, reset_rc; uint64_t bytes_in; part_t out; } job_t;

static void fetch(mdbg_ctx* c, void* dst, const void* src, uint64_t bytes) { if (bytes) CHECK(mdbg_copy_to_host(c, dst, src, bytes)); }

static void* rank_main(void* arg) {
    job_t* j = (job_t*)arg; world_t* w = j->rk.w;
    mdbg_comm comm; memset(&comm, 0, sizeof comm);          /* exchange_begin / exchange_wait stay NULL: this transport only has the blocking form */
    comm.self = &j->rk; comm.rank = j->rk.rank; comm.world = w->world;
    comm.allgather_u64 = t_allgather; comm.exchange = t_exchange; comm.allreduce_sum_u64 = t_allreduce;
    int err = 0;
    mdbg_dist* d = mdbg_dist_create(&j->P, &comm, &err);
    if (!d) { fprintf(stderr, "mdbg_dist_create: %d\n", err); exit(2); }
    if (j->chunks > 1) CHECK(mdbg_dist_set_pipeline(d, (uint32_t)j->chunks));      /* every ingest call below is cut into that many rounds */
    if (j->whole) CHECK(mdbg_dist_set_exchange(d, MDBG_EXCHANGE_WHOLE));            /* default: segments (only the hashes a peer's windows need) */
    mdbg_ctx* c = mdbg_dist_ctx(d);
    w->ctx[j->rk.rank] = c;
    pthread_barrier_wait(&w->bar);
    /* a second context generates this rank's reads (its buffers must outlive the rounds) */
    mdbg_ctx* gen = mdbg_create(&j->P, &err);
    const uint64_t per_round = j->reads_per_rank / (uint64_t)j->rounds;
    for (int rd = 0; rd < j->rounds; ++rd) {
        /* the last rank sits out the last round: every rank still takes part in the collective */
        const int idle = (rd == j->rounds - 1 && j->rk.rank == w->world - 1 && w->world > 1);
        const uint64_t first = ((uint64_t)rd * w->world + j->rk.rank) * per_round;       /* global ordinal of the batch's first read */
        mdbg_synth_params sp; memset(&sp, 0, sizeof sp);
        sp.seed = 7; sp.genome_len = j->genome; sp.n_reads = per_round; sp.mean_len = 9000; sp.sd_len = 1500; sp.min_len = 2000; sp.max_len = 16000; sp.err_ppm = 2000;
        const uint8_t* db = NULL; const uint64_t* dof = NULL; uint64_t nb = 0;
        CHECK(mdbg_synth_reads_device(gen, &sp, first, &db, &dof, &nb));
        if (idle) { CHECK(mdbg_dist_ingest_batch_device(d, NULL, NULL, 0, 0, 0)); continue; }
        if (j->packed) {
            uint64_t* words = NULL; uint64_t n_exc = 0;
            /* device scratch for the packed words: borrowed from a third context's synth buffer would be obscure - pack on the host instead */
            uint8_t* hb = (uint8_t*)malloc(nb); uint64_t* ho = (uint64_t*)malloc((per_round + 1) * 8);
            fetch(gen, hb, db, nb); fetch(gen, ho, dof, (per_round + 1) * 8);
            const uint64_t nw = (nb + 31) / 32;
            words = (uint64_t*)calloc(nw ? nw : 1, 8);
            for (uint64_t q = 0; q < nb; ++q) { words[q >> 5] |= (uint64_t)((hb[q] >> 1) & 1u) << (q & 31); words[q >> 5] |= (uint64_t)((hb[q] >> 2) & 1u) << (32 + (q & 31)); }
            /* device copies of words / offsets: reuse the generator's buffers (the ASCII bases are no longer needed) */
            CHECK(mdbg_copy_to_device(gen, (void*)db, words, nw * 8));
            mdbg_packed_batch pb; memset(&pb, 0, sizeof pb);
            pb.words = (const uint64_t*)db; pb.offsets = dof; pb.n_reads = per_round; pb.n_exc = n_exc;
            CHECK(mdbg_dist_ingest_batch_packed_device(d, &pb, nb, first));
            free(hb); free(ho); free(words);
        } else CHECK(mdbg_dist_ingest_batch_device(d, db, dof, per_round, nb, first));
    }
    mdbg_nodes nd; const uint64_t* d_row = NULL; uint64_t ng = 0;
    CHECK(mdbg_dist_finalize(d, &nd, &d_row, &ng));
    part_t* o = &j->out;
    o->n = nd.n; o->n_global = ng; o->n_distinct = nd.n_distinct;
    CHECK(mdbg_dist_traffic(d, &j->bytes_in, NULL, NULL));
    o->keys = (uint64_t*)malloc((nd.n * nd.k + 1) * 8); o->index = (uint32_t*)malloc((nd.n + 1) * 4); o->abundance = (uint16_t*)malloc((nd.n + 1) * 2);
    o->seqlen = (uint32_t*)malloc((nd.n + 1) * 4); o->src_read = (uint64_t*)malloc((nd.n + 1) * 8); o->row = (uint64_t*)malloc((nd.n + 1) * 8);
    o->src_start = (uint64_t*)malloc((nd.n + 1) * 8); o->src_end = (uint64_t*)malloc((nd.n + 1) * 8); o->shift_full = (uint64_t*)malloc((nd.n + 1) * 16);
    fetch(c, o->src_start, nd.src_start, nd.n * 8); fetch(c, o->src_end, nd.src_end, nd.n * 8); fetch(c, o->shift_full, nd.shift_full, nd.n * 16);
    fetch(c, o->keys, nd.keys, nd.n * nd.k * 8); fetch(c, o->index, nd.index, nd.n * 4); fetch(c, o->abundance, nd.abundance, nd.n * 2);
    fetch(c, o->seqlen, nd.seqlen, nd.n * 4); fetch(c, o->src_read, nd.src_read, nd.n * 8); fetch(c, o->row, d_row, nd.n * 8);
    pthread_barrier_wait(&w->bar);
    /* another k on the resident sketches: after a whole-sketch exchange every hash is here; after segments (which hold only this k's windows) the library runs the
       rounds' exchange again for the new k — collective either way */
    j->reset_rc = mdbg_dist_reset(d, j->P.k + 2);
    pthread_barrier_wait(&w->bar);
    mdbg_destroy(gen);
    mdbg_dist_destroy(d);
    return NULL;
}

int main(int argc, char** argv) {
    const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 2;
    const uint64_t rpr = argc > 2 ? strtoull(argv[2], NULL, 10) : 300;
    const int rounds = argc > 3 ? atoi(argv[3]) : 2;
    const int packed = argc > 4 ? atoi(argv[4]) : 0;
    const int chunks = argc > 5 ? atoi(argv[5]) : 1;          /* > 1: mdbg_dist_set_pipeline */
    const int whole = argc > 6 ? atoi(argv[6]) : 0;           /* 1: MDBG_EXCHANGE_WHOLE */
    if (chunks < 1 || chunks > 64) { fprintf(stderr, "bad arguments\n"); return 1; }
    if (W < 1 || W > MAXW || rounds < 1 || rpr % (uint64_t)rounds) { fprintf(stderr, "bad arguments\n"); return 1; }
    mdbg_params P; memset(&P, 0, sizeof P);
    P.k = 9; P.l = 12; P.density = 0.004; P.min_abundance = 2; P.device = -1;
    world_t w; memset(&w, 0, sizeof w); w.world = W;
    pthread_barrier_init(&w.bar, NULL, W);
    job_t* jobs = (job_t*)calloc(W, sizeof(job_t));
    pthread_t th[MAXW];
    const uint64_t genome = 150000;
    for (uint32_t r = 0; r < W; ++r) { jobs[r].rk.w = &w; jobs[r].rk.rank = r; jobs[r].P = P; jobs[r].reads_per_rank = rpr; jobs[r].genome = genome; jobs[r].rounds = rounds; jobs[r].packed = packed; jobs[r].chunks = chunks; jobs[r].whole = whole; }
    for (uint32_t r = 0; r < W; ++r) pthread_create(&th[r], NULL, rank_main, &jobs[r]);
    for (uint32_t r = 0; r < W; ++r) pthread_join(th[r], NULL);

    /* reference: ONE context over the same reads (same ordinals) */
    int err = 0;
    mdbg_ctx* one = mdbg_create(&P, &err), *gen = mdbg_create(&P, &err);
    const uint64_t per_round = rpr / (uint64_t)rounds;
    for (int rd = 0; rd < rounds; ++rd) for (uint32_t r = 0; r < W; ++r) {
        if (rd == rounds - 1 && r == W - 1 && W > 1) continue;
        const uint64_t first = ((uint64_t)rd * W + r) * per_round;
        mdbg_synth_params sp; memset(&sp, 0, sizeof sp);
        sp.seed = 7; sp.genome_len = genome; sp.n_reads = per_round; sp.mean_len = 9000; sp.sd_len = 1500; sp.min_len = 2000; sp.max_len = 16000; sp.err_ppm = 2000;
        const uint8_t* db; const uint64_t* dof; uint64_t nb;
        CHECK(mdbg_synth_reads_device(gen, &sp, first, &db, &dof, &nb));
        CHECK(mdbg_ingest_batch_device(one, db, dof, per_round, nb, first));
    }
    mdbg_nodes ref;
    CHECK(mdbg_finalize(one, &ref));
    uint64_t total = 0;
    for (uint32_t r = 0; r < W; ++r) total += jobs[r].out.n;
    int ok = total == ref.n && ref.n > 100;
    unsigned char* seen = (unsigned char*)calloc(ref.n + 1, 1);
    for (uint32_t r = 0; r < W && ok; ++r) {
        const part_t* o = &jobs[r].out;
        ok = ok && o->n_global == ref.n && o->n_distinct == ref.n_distinct;
        for (uint64_t i = 0; i < o->n && ok; ++i) {
            const uint64_t row = o->row[i];
            ok = row < ref.n && !seen[row] && o->index[i] == ref.index[row] && o->abundance[i] == ref.abundance[row] && o->seqlen[i] == ref.seqlen[row] &&
                 o->src_read[i] == ref.src_read[row] && !memcmp(o->keys + i * ref.k, ref.keys + row * ref.k, ref.k * 8) &&
                 /* what needs the raw positions of the A-th sighting — fetched from the rank that sketched the read */
                 o->src_start[i] == ref.src_start[row] && o->src_end[i] == ref.src_end[row] && !memcmp(o->shift_full + 2 * i, ref.shift_full + 2 * row, 16);
            if (row < ref.n) seen[row] = 1;
        }
    }
    printf("world %u, %llu reads per rank in %d rounds (x %d pipelined chunks), %s input: %llu nodes (%llu distinct k-min-mers); partitions", W, (unsigned long long)rpr, rounds, chunks,
           packed ? "packed" : "ASCII", (unsigned long long)ref.n, (unsigned long long)ref.n_distinct);
    for (uint32_t r = 0; r < W; ++r) printf(" %llu", (unsigned long long)jobs[r].out.n);
    { uint64_t bi = 0; for (uint32_t r = 0; r < W; ++r) bi += jobs[r].bytes_in; printf("; %s exchange, bytes received by all ranks: %llu", whole ? "whole-sketch" : "segment", (unsigned long long)bi); }
    { int same = 1; for (uint32_t r = 1; r < W; ++r) same = same && jobs[r].reset_rc == jobs[0].reset_rc; printf("; reset(k + 2): %d%s", jobs[0].reset_rc, same ? "" : " (ranks differ)"); }
    printf(" -> %s\n", ok ? "EQUAL to the single-context table" : "MISMATCH");
    mdbg_destroy(one); mdbg_destroy(gen);
    return ok ? 0 : 4;
}
