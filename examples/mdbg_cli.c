/* mdbg_cli.c — the two C ABIs used from plain C, the way rust-mdbg's main() would use them through FFI:
 *   reads.fa[.gz]  ->  mdbg_reader_*  ->  mdbg_ingest_batch  ->  mdbg_finalize  ->  mdbg_graph_edges  ->  <prefix>.gfa (+ <prefix>.0.sequences)
 * Same flags as the reference binary for this path (src/main.rs:330-420): -k -l --density --minabund --presimp --prefix.
 * Build:  gcc -O2 -Iinclude examples/mdbg_cli.c -Lrust_mdbg_amd -lmdbg_hip -lmdbg_emit -Wl,-rpath,$PWD/rust_mdbg_amd -o mdbg_cli
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdbg_emit.h"
#include "mdbg_hip.h"

static void die(mdbg_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, mdbg_strerror(rc), ctx && mdbg_last_error(ctx) ? mdbg_last_error(ctx) : "");
    exit(1);
}

int main(int argc, char** argv) {
    mdbg_params p; memset(&p, 0, sizeof p);
    p.k = 10; p.l = 12; p.density = 0.1; p.min_abundance = 2; p.device = -1;       /* the reference's defaults (main.rs:430-450) */
    float presimp = 0.01f;
    const char* input = NULL; const char* prefix = "graph"; int write_sequences = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-k") && i + 1 < argc) p.k = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) p.l = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--density") && i + 1 < argc) p.density = atof(argv[++i]);
        else if (!strcmp(argv[i], "--minabund") && i + 1 < argc) p.min_abundance = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--presimp") && i + 1 < argc) presimp = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--prefix") && i + 1 < argc) prefix = argv[++i];
        else if (!strcmp(argv[i], "--no-basespace")) write_sequences = 0;
        else if (argv[i][0] != '-') input = argv[i];
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!input) { fprintf(stderr, "usage: mdbg_cli reads.fa[.gz] [-k K] [-l L] [--density D] [--minabund A] [--presimp P] [--prefix PFX] [--no-basespace]\n"); return 2; }

    int err = 0;
    mdbg_ctx* ctx = mdbg_create(&p, &err);
    if (!ctx) die(NULL, "mdbg_create", err);
    mdbg_reader* rd = mdbg_reader_open(input, 0, &err);
    if (!rd) die(NULL, "mdbg_reader_open", err);
    uint64_t n_reads = 0, first = 0;
    for (;;) {
        const uint8_t* bases; const uint64_t* offs; uint64_t n;
        int rc = mdbg_reader_next(rd, 256u << 20, &bases, &offs, &n);
        if (rc) die(NULL, "mdbg_reader_next", rc);
        if (!n) break;
        rc = mdbg_ingest_batch(ctx, bases, offs, n, first);          /* process_read_aux over the batch */
        if (rc) die(ctx, "mdbg_ingest_batch", rc);
        first += n; n_reads += n;
    }
    mdbg_reader_close(rd);

    mdbg_nodes nodes; mdbg_edge_list edges;
    int rc = mdbg_finalize(ctx, &nodes);
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
    if (write_sequences) {                                          /* second pass over the input: the node sequences */
        snprintf(path, sizeof path, "%s.0.sequences", prefix);
        mdbg_seqfile* sf = mdbg_seqfile_open(path, p.k, p.l, &err);
        if (!sf) die(NULL, "mdbg_seqfile_open", err);
        rd = mdbg_reader_open(input, 0, &err);
        if (!rd) die(NULL, "mdbg_reader_open", err);
        first = 0;
        for (;;) {
            const uint8_t* bases; const uint64_t* offs; uint64_t n;
            rc = mdbg_reader_next(rd, 256u << 20, &bases, &offs, &n);
            if (rc) die(NULL, "mdbg_reader_next", rc);
            if (!n) break;
            rc = mdbg_seqfile_write_batch(sf, &nodes, bases, offs, n, first);
            if (rc) die(NULL, "mdbg_seqfile_write_batch", rc);
            first += n;
        }
        mdbg_reader_close(rd);
        rc = mdbg_seqfile_close(sf);
        if (rc) die(NULL, "mdbg_seqfile_close", rc);
    }
    mdbg_destroy(ctx);
    return 0;
}
