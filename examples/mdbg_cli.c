/* mdbg_cli.c — the two C ABIs used from plain C, the way rust-mdbg's main() would use them through FFI:
 *   reads.fa[.gz]  ->  mdbg_reader_*  ->  mdbg_ingest_batch[_packed]  ->  mdbg_finalize  ->  mdbg_graph_edges  ->  <prefix>.gfa (+ <prefix>.0.sequences)
 * Same flags as the reference binary for this path (src/main.rs:330-420): -k -l --density --minabund --presimp --prefix --threads
 * --reference --skiphpc --syncmers/-s --lmer-counts/--lmer_counts_min/--lmer_counts_max --no-basespace.
 * --threads N > 1: an uncompressed input is mapped and parsed by N threads (mdbg_reader_open_mt) that also pack their pieces to 2 bits
 * per base (mdbg_reader_next_packed); a reader thread produces batch i+1 while the main thread ingests batch i.
 * Build:  gcc -O2 -Iinclude examples/mdbg_cli.c -Lrust_mdbg_amd -lmdbg_hip -lmdbg_emit -lpthread -Wl,-rpath,$PWD/rust_mdbg_amd -o mdbg_cli
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mdbg_emit.h"
#include "mdbg_hip.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* reader thread: one batch ahead of the consumer.  The reader alternates two buffer sets (a batch stays valid until the call after
 * the next one), so exactly one batch may be outstanding while the next is parsed and packed. */
typedef struct feed_t {
    mdbg_reader* rd; uint64_t max_bases;
    pthread_mutex_t mu; pthread_cond_t cv;
    int have, rc;                          /* have: a batch is waiting to be picked up */
    mdbg_packed_batch pb;
} feed_t;
static void* feed_main(void* arg) {
    feed_t* f = (feed_t*)arg;
    for (;;) {
        mdbg_packed_batch pb;
        const int rc = mdbg_reader_next_packed(f->rd, f->max_bases, &pb);
        const uint64_t n = rc ? 0 : pb.n_reads;
        pthread_mutex_lock(&f->mu);
        f->pb = pb; f->rc = rc; f->have = 1;
        pthread_cond_broadcast(&f->cv);
        while (f->have) pthread_cond_wait(&f->cv, &f->mu);          /* picking batch j up means the consumer is done with batch j-1: its buffer may be reused */
        pthread_mutex_unlock(&f->mu);
        if (rc || !n) return NULL;
    }
}
/* next batch from the reader thread (the previous one must have been fully consumed) */
static int feed_take(feed_t* f, mdbg_packed_batch* pb) {
    pthread_mutex_lock(&f->mu);
    while (!f->have) pthread_cond_wait(&f->cv, &f->mu);
    *pb = f->pb;
    const int rc = f->rc;
    f->have = 0;
    pthread_cond_broadcast(&f->cv);
    pthread_mutex_unlock(&f->mu);
    return rc;
}

/* one writer of the .sequences pass: the lines of the nodes i with i % n_parts == part of the current batch, into its own file */
typedef struct seqjob_t { mdbg_seqfile* sf; const mdbg_nodes* nodes; uint32_t part, n_parts; const uint8_t* bases; const uint64_t* offs; uint64_t n, first; int rc; } seqjob_t;
static void* seqjob_main(void* arg) {
    seqjob_t* j = (seqjob_t*)arg;
    j->rc = mdbg_seqfile_write_batch_part(j->sf, j->nodes, j->part, j->n_parts, j->bases, j->offs, j->n, j->first);
    return NULL;
}

static void die(mdbg_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, mdbg_strerror(rc), ctx && mdbg_last_error(ctx) ? mdbg_last_error(ctx) : "");
    exit(1);
}

int main(int argc, char** argv) {
    mdbg_params p; memset(&p, 0, sizeof p);
    p.k = 10; p.l = 12; p.density = 0.1; p.min_abundance = 2; p.device = -1;       /* the reference's defaults (main.rs:430-450) */
    float presimp = 0.01f;
    const char* input = NULL; const char* prefix = "graph"; int write_sequences = 1, threads = 1, reference = 0, timing = 0;
    const char* lmer_counts = NULL; uint32_t lc_min = 2, lc_max = 100000;          /* main.rs:447-448 */
    int syncmer_s_given = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-k") && i + 1 < argc) p.k = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) p.l = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--density") && i + 1 < argc) p.density = atof(argv[++i]);
        else if (!strcmp(argv[i], "--minabund") && i + 1 < argc) p.min_abundance = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--presimp") && i + 1 < argc) presimp = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--prefix") && i + 1 < argc) prefix = argv[++i];
        else if (!strcmp(argv[i], "--no-basespace")) write_sequences = 0;
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reference")) reference = 1;                                 /* main.rs:737: newlines inside FASTA records are removed */
        else if (!strcmp(argv[i], "--lmer-counts") && i + 1 < argc) lmer_counts = argv[++i];
        else if (!strcmp(argv[i], "--lmer_counts_min") && i + 1 < argc) lc_min = (uint32_t)strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--lmer_counts_max") && i + 1 < argc) lc_max = (uint32_t)strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--timing")) timing = 1;
        else if (!strcmp(argv[i], "--skiphpc")) p.reads_already_hpc = 1;                         /* main.rs:490 */
        else if (!strcmp(argv[i], "--syncmers")) { p.scheme = MDBG_SCHEME_SYNCMERS; if (!syncmer_s_given) p.syncmer_s = 4; }       /* main.rs:438,491-495: default s = 4 */
        else if ((!strcmp(argv[i], "-s") || !strcmp(argv[i], "--s")) && i + 1 < argc) { p.syncmer_s = (uint32_t)atoi(argv[++i]); syncmer_s_given = 1; }
        else if (argv[i][0] != '-') input = argv[i];
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!input) { fprintf(stderr, "usage: mdbg_cli reads.fa[.gz] [-k K] [-l L] [--density D] [--minabund A] [--presimp P] [--prefix PFX] [--no-basespace] [--threads N] [--reference] [--skiphpc] [--syncmers [-s S]] [--lmer-counts FILE [--lmer_counts_min A] [--lmer_counts_max B]] [--timing]\n"); return 2; }
    if (threads < 1) threads = 1;

    int err = 0;
    mdbg_ctx* ctx = mdbg_create(&p, &err);
    if (!ctx) die(NULL, "mdbg_create", err);
    const double t0 = now_s();
    if (lmer_counts) {                                               /* main.rs:544-575: the selected l-mers restrict the sketch */
        uint64_t* codes = NULL; uint64_t n_codes = 0, ignored = 0;
        int rc = mdbg_lmer_filter_from_counts(lmer_counts, p.l, p.density, lc_min, lc_max, &codes, &n_codes, &ignored);
        if (rc) die(NULL, "mdbg_lmer_filter_from_counts", rc);
        rc = mdbg_set_lmer_filter(ctx, codes, n_codes);
        if (rc) die(ctx, "mdbg_set_lmer_filter", rc);
        mdbg_lmer_filter_free(codes);
    }
    const uint64_t batch_bases = 256u << 20;
    mdbg_reader* rd = mdbg_reader_open_mt(input, reference, threads, &err);
    if (!rd) die(NULL, "mdbg_reader_open", err);
    /* the batch buffers come from the GPU library: page-locked by the first ingest call that sees them, so a batch crosses PCIe as one DMA */
    { const int rc = mdbg_reader_set_allocator(rd, mdbg_host_alloc, mdbg_host_free); if (rc) die(NULL, "mdbg_reader_set_allocator", rc); }
    uint64_t n_reads = 0, n_bases = 0, first = 0;
    double t_wait = 0, t_gpu = 0;                                   /* --timing: where the ingest loop spends its time */
    if (threads > 1) {                                              /* any input: the streaming reader (.gz, .lz4) packs on the reader thread */
        /* reader thread one batch ahead; this thread packs to 2 bits per base and ingests */
        feed_t f; memset(&f, 0, sizeof f); f.rd = rd; f.max_bases = batch_bases;
        pthread_mutex_init(&f.mu, NULL); pthread_cond_init(&f.cv, NULL);
        pthread_t th; pthread_create(&th, NULL, feed_main, &f);
        for (;;) {
            mdbg_packed_batch pb;
            double ta = now_s();
            int rc = feed_take(&f, &pb);
            t_wait += now_s() - ta;
            if (rc) die(NULL, "mdbg_reader_next_packed", rc);
            if (!pb.n_reads) break;
            ta = now_s();
            rc = mdbg_ingest_batch_packed(ctx, &pb, first);
            if (rc) die(ctx, "mdbg_ingest_batch_packed", rc);
            t_gpu += now_s() - ta;
            first += pb.n_reads; n_reads += pb.n_reads; n_bases += pb.offsets[pb.n_reads];
        }
        pthread_join(th, NULL);
    } else for (;;) {
        const uint8_t* bases; const uint64_t* offs; uint64_t n;
        int rc = mdbg_reader_next(rd, batch_bases, &bases, &offs, &n);
        if (rc) die(NULL, "mdbg_reader_next", rc);
        if (!n) break;
        rc = mdbg_ingest_batch(ctx, bases, offs, n, first);          /* process_read_aux over the batch */
        if (rc) die(ctx, "mdbg_ingest_batch", rc);
        first += n; n_reads += n; n_bases += offs[n];
    }
    mdbg_reader_close(rd);
    const double t_ingest = now_s();

    mdbg_nodes nodes; mdbg_edge_list edges;
    /* without the .sequences pass the host prints three columns of the node table (the S lines): the minimizer lists stay on the device */
    int rc = write_sequences ? mdbg_finalize(ctx, &nodes) : mdbg_finalize_gfa(ctx, &nodes);
    if (rc) die(ctx, "mdbg_finalize", rc);
    rc = mdbg_graph_edges(ctx, presimp, &edges);
    if (rc) die(ctx, "mdbg_graph_edges", rc);
    if (p.min_abundance > 1) {                                      /* what the reference prints (main.rs:926-928, 1118-1120) */
        printf("Number of nodes before abundance filter: %llu\n", (unsigned long long)nodes.n_distinct);
        printf("Number of nodes after abundance filter: %llu\n", (unsigned long long)nodes.n);
    } else printf("Number of mdBG nodes: %llu\n", (unsigned long long)nodes.n);
    printf("Number of mdBG edges: %llu\n", (unsigned long long)edges.n);
    printf("Pre-simp = %g: %llu edges removed\n", presimp, (unsigned long long)edges.presimp_removed);

    char path[4096];
    snprintf(path, sizeof path, "%s.gfa", prefix);
    rc = mdbg_emit_write_gfa(path, &nodes, &edges);
    if (rc) die(NULL, "mdbg_emit_write_gfa", rc);
    if (timing) fprintf(stderr, "timing: %llu reads, %llu bases; ingest %.3f s, to .gfa %.3f s (%.2f Gbases/s; context creation not included); ingest loop: waiting for the reader %.3f, mdbg_ingest_batch_packed %.3f s\n",
                        (unsigned long long)n_reads, (unsigned long long)n_bases, t_ingest - t0, now_s() - t0, (double)n_bases / (now_s() - t0) / 1e9, t_wait, t_gpu);
    if (write_sequences) {                                          /* second pass over the input: the node sequences */
        /* one file per writer thread, "<prefix>.<t>.sequences", as the reference's worker threads write them (main.rs:614-630) */
        enum { MAX_WRITERS = 16 };
        const int nw = threads > MAX_WRITERS ? MAX_WRITERS : threads;
        mdbg_seqfile* sf[MAX_WRITERS]; seqjob_t job[MAX_WRITERS]; pthread_t th[MAX_WRITERS];
        for (int t = 0; t < nw; ++t) {
            snprintf(path, sizeof path, "%s.%d.sequences", prefix, t);
            sf[t] = mdbg_seqfile_open(path, p.k, p.l, &err);
            if (!sf[t]) die(NULL, "mdbg_seqfile_open", err);
        }
        rd = mdbg_reader_open_mt(input, reference, threads, &err);
        if (!rd) die(NULL, "mdbg_reader_open", err);
        first = 0;
        const double ts = now_s();
        for (;;) {
            const uint8_t* bases; const uint64_t* offs; uint64_t n;
            rc = mdbg_reader_next(rd, 256u << 20, &bases, &offs, &n);
            if (rc) die(NULL, "mdbg_reader_next", rc);
            if (!n) break;
            for (int t = 0; t < nw; ++t) {
                seqjob_t jb; jb.sf = sf[t]; jb.nodes = &nodes; jb.part = (uint32_t)t; jb.n_parts = (uint32_t)nw; jb.bases = bases; jb.offs = offs; jb.n = n; jb.first = first; jb.rc = 0;
                job[t] = jb;
                if (t) pthread_create(&th[t], NULL, seqjob_main, &job[t]);
            }
            seqjob_main(&job[0]);
            for (int t = 1; t < nw; ++t) pthread_join(th[t], NULL);
            for (int t = 0; t < nw; ++t) if (job[t].rc) die(NULL, "mdbg_seqfile_write_batch_part", job[t].rc);
            first += n;
        }
        mdbg_reader_close(rd);
        for (int t = 0; t < nw; ++t) { rc = mdbg_seqfile_close(sf[t]); if (rc) die(NULL, "mdbg_seqfile_close", rc); }
        if (timing) fprintf(stderr, "timing: .sequences pass %.3f s (%d file%s)\n", now_s() - ts, nw, nw > 1 ? "s" : "");
    }
    mdbg_destroy(ctx);
    return 0;
}
