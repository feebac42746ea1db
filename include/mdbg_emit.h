/* mdbg_emit.h — host-side graph emitter above the node table of mdbg_hip.h (plain C ABI, no GPU involved).
 *
 * Mirrors what rust-mdbg's main() does AFTER the hot path with the read-only node view (paths in the rust-mdbg tree):
 *   mdbg_emit_edges            (k-1)-mer index, 4-orientation overlap test, presimp            src/main.rs:1014-1117
 *   mdbg_emit_write_gfa        "H\tVN:Z:1.0", S-lines, L-lines                                 src/main.rs:1011,1021,1095,1113
 *   mdbg_seqfile_*             {prefix}.{tid}.sequences: header + one line per solid node,     src/main.rs:614-630,693-708
 *                              LZ4 frame (stored blocks) readable by src/to_basespace.rs:62-66,203-241
 * The reference iterates its DashMap in arbitrary order; here nodes are visited in index order, so files are deterministic.
 */
#ifndef MDBG_EMIT_H
#define MDBG_EMIT_H

#include <stddef.h>
#include <stdint.h>

#include "mdbg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdbg_emit mdbg_emit; /* owns the edge buffers it hands out */

/* n = number of L-lines ("Number of mdBG edges", main.rs:1118); n1/n2 = DbgEntry.index; o1/o2 = '+' or '-';
 * overlap = min(n1.seqlen - shift, n2.seqlen - 1) (main.rs:1091-1092); presimp_removed: main.rs:1120.
 * Same type as the GPU edge list of include/mdbg_hip.h (mdbg_graph_edges). */
typedef mdbg_edge_list mdbg_edges;

mdbg_emit* mdbg_emit_create(void);
void mdbg_emit_destroy(mdbg_emit* e);
/* nodes: HOST arrays (as returned by mdbg_finalize).  presimp: --presimp, reference default 0.01 (main.rs:449). */
int mdbg_emit_edges(mdbg_emit* e, const mdbg_nodes* nodes, float presimp, mdbg_edges* out);
int mdbg_emit_write_gfa(const char* path, const mdbg_nodes* nodes, const mdbg_edges* edges);

/* .sequences writer.  Open once, feed every batch of reads that was ingested (same buffers / ordinals as
 * mdbg_ingest_batch): the nodes whose A-th sighting lies in that batch get their line.  Close to finish the frame. */
typedef struct mdbg_seqfile mdbg_seqfile;
mdbg_seqfile* mdbg_seqfile_open(const char* path, uint32_t k, uint32_t l, int* err);
int mdbg_seqfile_write_batch(mdbg_seqfile* f, const mdbg_nodes* nodes, const uint8_t* bases, const uint64_t* offsets,
                             uint64_t n_reads, uint64_t first_read_ordinal);
/* The same for the nodes i with i % n_parts == part only: the reference writes one `.sequences` file per worker thread
 * ("{prefix}.{thread}.sequences", src/main.rs:614-630) and its readers take all of them; n_parts files written by n_parts threads, each
 * calling this with its own file and part on the same batch, give the same set of lines in parallel. */
int mdbg_seqfile_write_batch_part(mdbg_seqfile* f, const mdbg_nodes* nodes, uint32_t part, uint32_t n_parts, const uint8_t* bases,
                                  const uint64_t* offsets, uint64_t n_reads, uint64_t first_read_ordinal);
int mdbg_seqfile_close(mdbg_seqfile* f);

/* ---- host ingest: FASTA / FASTQ, optionally gzip-compressed, into the batch layout of
 * mdbg_ingest_batch.  Mirrors get_reader + the seq_io readers of the reference (src/main.rs:163-178,461-467,830-839):
 * the format is decided by the FILE NAME (".fa"/".fasta" suffix or ".fa."/".fasta." inside -> FASTA, anything else ->
 * FASTQ); a regular file that starts with the gzip magic is mapped and inflated by the library's own decoder (csrc/gz_inflate.h: concatenated
 * members like the reference's MultiGzDecoder, CRC-32 and length of every member checked, ~3x zlib's rate; a damaged or truncated stream makes
 * mdbg_reader_next return MDBG_E_IO), other input goes through zlib's gzread (which passes plain text through), ".lz4" through a built-in LZ4
 * frame decoder (the image has no liblz4).  Like seq_io's RefRecord::seq(), a multi-line
 * FASTA record keeps its interior line terminators unless strip_newlines is set (the reference strips them only with
 * --reference, src/main.rs:737; otherwise such a read trips the ACGTN check, as it does in the reference). */
typedef struct mdbg_reader mdbg_reader;
mdbg_reader* mdbg_reader_open(const char* path, int strip_newlines, int* err);
/* Next batch of whole records, at most max_bases bases unless a single record is longer (then that record alone).
 * *n_reads == 0 at end of file.  bases/offsets (n_reads + 1 entries, offsets[0] = 0) belong to the reader and stay valid
 * until the next call on it. */
int mdbg_reader_next(mdbg_reader* r, uint64_t max_bases, const uint8_t** bases, const uint64_t** offsets, uint64_t* n_reads);
int mdbg_reader_is_fasta(const mdbg_reader* r);
/* The same reader with `threads` parser threads for UNCOMPRESSED files (seq_io's parallel readers, src/main.rs:830-839, parse records on
 * one thread and only run the per-read work in parallel): every batch is a window of about max_bases file bytes (2 * max_bases for FASTQ) cut at
 * record starts, so batches hold the same records in the same order with the same bytes as mdbg_reader_open's, only cut at other places (never
 * more than max_bases bases unless a window's last record is longer).  A window whose records are "header line + one sequence line" (FASTA) or
 * four-line records (FASTQ, as the streaming reader takes them) is read ONCE: in chunks of 256 KB that up to 24 of the threads claim in file
 * order, each read with pread into the thread's own buffer, its records located there and copied / packed from there to their place in the
 * batch (a running sum of bases and records handed down the chunks).  Any other window (FASTA sequences over several lines, strip_newlines,
 * stray lines, a record of many megabytes) is cut into one piece per thread and parsed by the streaming reader's record code on the mapped file.
 * .lz4 input and threads <= 1 fall back to the streaming reader.  gzip input is inflated AHEAD of the parser on a thread of its own — a BGZF file
 * (bgzip: independent blocks of at most 64 KiB) by several threads at once, an ordinary gzip stream by one — and the inflated text is parsed in
 * windows the same way: the part of a window that holds whole records goes to the parser threads, the rest is carried into the next window.
 * When the file is read in parallel, mdbg_reader_next alternates two buffers: a batch stays valid until the call AFTER the next one, so that
 * another thread can pack / copy / ingest batch i while batch i+1 is being read. */
mdbg_reader* mdbg_reader_open_mt(const char* path, int strip_newlines, int threads, int* err);
/* Where the batch buffers handed out by a parallel reader (mdbg_reader_open_mt on an uncompressed file: the ASCII bases of mdbg_reader_next, the packed words of
 * mdbg_reader_next_packed of any reader) come from; default malloc / free.  With mdbg_host_alloc / mdbg_host_free of mdbg_hip.h the ingest calls page-lock them on
 * first use and the copy to the device is one DMA.  Before the first batch (MDBG_E_STATE afterwards); both or neither (NULL, NULL = malloc / free). */
int mdbg_reader_set_allocator(mdbg_reader* r, void* (*alloc_fn)(size_t), void (*free_fn)(void*));
int mdbg_reader_is_parallel(const mdbg_reader* r);      /* 1: the input is read by several threads (two alternating batch buffers) */
void mdbg_reader_close(mdbg_reader* r);

/* The next batch straight in the 2-bit packed layout (mdbg_packed_batch of mdbg_hip.h, HOST memory owned by the reader): what
 * mdbg_reader_next + mdbg_pack_reads would give, without the ASCII copy of the batch in between — with a parallel reader every parser
 * thread packs its own piece.  out->n_reads == 0 at end of file; the number of bases is out->offsets[out->n_reads].  The arrays stay valid
 * until the call AFTER the next one (two alternating sets), for every kind of reader.  Do not mix with mdbg_reader_next on one reader. */
int mdbg_reader_next_packed(mdbg_reader* r, uint64_t max_bases, mdbg_packed_batch* out);

/* ---- host packer for mdbg_ingest_batch_packed (layout: mdbg_packed_batch in mdbg_hip.h) -----------------------------
 * Replaces the per-read String copy of src/main.rs:733-739: the reader's ASCII batch is squeezed to 2 bits per base
 * (two 32-bit planes per 32 bases) before it crosses PCIe.  words: (n_bases + 31) / 32 entries.  Bytes outside "ACGT" go
 * to exc_pos / exc_val (ascending; room for exc_cap entries); *n_exc is their number, MDBG_E_CAPACITY if it exceeds exc_cap
 * (the words are complete either way).  threads <= 1: the calling thread only. */
/* --lmer-counts FILE --lmer_counts_min A --lmer_counts_max B (src/main.rs:392-409, 544-575; src/minimizers.rs:53-113): reads the counts
 * file ("lmer count" per line, as kmc_dump writes it), keys every l-mer by min(lmer, revcomp), and SELECTS an l-mer iff its count is
 * inside (count_min, count_max) exclusive and f64(canonical ntHash) / 2^64 <= density.  *codes (malloc'ed, free with
 * mdbg_lmer_filter_free; sorted, unique) lists the selected l-mers AND their reverse complements as 2-bit codes in the layout
 * mdbg_set_lmer_filter takes.  Lines whose l-mer has another length than l or a byte outside ACGT cannot match any read l-mer of the
 * density path and are counted in *n_ignored (may be NULL).  Defaults of the reference: count_min 2, count_max 100000.  l <= 32.
 * MDBG_E_IO: the file cannot be read; MDBG_E_PARAM: a line without a valid u32 count (the reference panics). */
int mdbg_lmer_filter_from_counts(const char* path, uint32_t l, double density, uint32_t count_min, uint32_t count_max,
                                 uint64_t** codes, uint64_t* n_codes, uint64_t* n_ignored);
void mdbg_lmer_filter_free(uint64_t* codes);

uint64_t mdbg_packed_words(uint64_t n_bases);
int mdbg_pack_reads(const uint8_t* bases, uint64_t n_bases, uint64_t* words, uint64_t* exc_pos, uint8_t* exc_val,
                    uint64_t exc_cap, uint64_t* n_exc, int threads);

#ifdef __cplusplus
}
#endif
#endif
