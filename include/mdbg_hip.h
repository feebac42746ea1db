/* mdbg_hip.h — C ABI of libmdbg_hip.so: MI355X-native minimizer sketching + k-min-mer counting.
 *
 * This is the drop-in boundary for rust-mdbg's per-read hot path.  The reference has NO plugin/FFI
 * interface for it (the path is a set of closures inside fn main()), so the boundary is cut along the
 * seams SURVEY.md §8b identifies; each entry point names the reference code it replaces
 * (paths relative to the rust-mdbg tree):
 *
 *   mdbg_create            Params {k,l,density,min_kmer_abundance,reads_already_hpc}   src/main.rs:92-114,515-537
 *                          + dbg_nodes / NODE_INDEX construction                       src/main.rs:595-598
 *   mdbg_ingest_batch      process_read_aux over a batch of records                    src/main.rs:730-785
 *     (= Read::extract_density  src/read.rs:176-211, encode_rle src/read.rs:157-174, nthash::NtHashIterator,
 *        the k-min-mer window loop src/main.rs:756-781, KmerVec::normalize src/kmer_vec.rs:34-39 and
 *        add_kminmer's counting upsert src/main.rs:632-691)
 *   mdbg_ingest_batch_packed   the same over reads packed 2 bits per base by the host (mdbg_pack_reads of mdbg_emit.h), replacing
 *                          the per-read String copies of src/main.rs:733-739: a quarter of the bytes over PCIe and through HBM
 *   mdbg_sketch_only       Read::extract (density scheme)                              src/read.rs:85-90,176-211
 *   mdbg_finalize          abundance filter + read-only view of dbg_nodes              src/main.rs:922-929,1014-1016
 *                          + what the .sequences line of a node is built from          src/main.rs:693-708
 *   mdbg_graph_edges       km_index + orientation tests + presimp + overlaps           src/main.rs:1017-1117
 *   mdbg_query_batch       --read_stats: abundance of every k-min-mer of a query read  src/main.rs:939-1004
 *   mdbg_reset             a new k over the same reads (utils/multik:69-78 re-runs the binary per k)
 *   mdbg_set_lmer_filter   --lmer-counts: restrict the minimizers to the l-mers selected from a counts file
 *   mdbg_mark / _rewind    ... with the script's contig feedback: forget the last round's contigs, keep the reads' sketches
 *   mdbg_destroy           process exit
 *
 * Conventions: every function returns 0 (MDBG_OK) or a negative error code and never aborts (the reference
 * panics); strings are (ptr,len); the caller owns its input buffers and may free them when a call returns;
 * the library owns the context and every buffer it hands out until the next call that documents otherwise
 * or mdbg_destroy.  Semantics are those of the reference run with `--threads 1` and without `--bf`:
 * results do not depend on how reads are split into batches or on the order of the calls, only on
 * `first_read_ordinal` (the position of the batch's first record in the input file).
 *
 * No torch / HIP types appear here: device buffers are plain pointers.
 */
#ifndef MDBG_HIP_H
#define MDBG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDBG_ABI_VERSION 3      /* 3: the owner of a k-min-mer is a function of its smallest hash (ranks of a multi-GPU job must run the same version) */

enum {
    MDBG_OK = 0,
    MDBG_E_PARAM = -1,    /* bad parameter (k<2, l out of range, minabund==0, null pointer ...) */
    MDBG_E_ALPHABET = -2, /* a read whose HPC length is >= l holds a byte outside "ACGTN" (reference: nthash panics) */
    MDBG_E_CAPACITY = -3, /* a fixed limit was exceeded (e.g. more than 2^26 minimizers in one read) */
    MDBG_E_DEVICE = -4,   /* HIP runtime failure; text via mdbg_last_error */
    MDBG_E_NOMEM = -5,    /* device or host allocation failed */
    MDBG_E_STATE = -6,    /* call not valid in the context's current state (e.g. ingest after an error) */
    MDBG_E_IO = -7        /* libmdbg_emit: a file could not be opened, or a write / close failed (disk full ...) */
};

#define MDBG_MAX_L 255u        /* l = 2..32 run on the bit-sliced kernel, longer l-mers on its generic exact walker (reference: unbounded) */
#define MDBG_MAX_MINABUND 65535u /* DbgAbundance is a u16 in the reference; up to 8 the table tracks the A-th sighting directly, above it is recovered at finalize */
#define MDBG_FLAG_FORCE_GENERIC 1u /* every tile takes the generic exact sketch kernel (testing / cross-check) */
#define MDBG_SCHEME_DENSITY 0u   /* canonical ntHash <= density * 2^64           Read::extract_density   src/read.rs:176-211 */
#define MDBG_SCHEME_SYNCMERS 1u  /* --syncmers: open syncmers (smallest s-mer in the middle), down-sampled by hash(l-mer) <= density * 4^l;
                                  * 2-bit codes, A/a C/c G/g T/t/U/u, any other byte resets (no alphabet error); l <= 31
                                  *                                              Read::extract_syncmers  src/read.rs:215-352 */

typedef struct mdbg_ctx mdbg_ctx;

/* src/main.rs:92-114 (the fields this path reads) */
typedef struct mdbg_params {
    uint32_t k;                 /* k-min-mer length, >= 2                                   (-k) */
    uint32_t l;                 /* minimizer length, 2..MDBG_MAX_L                          (-l) */
    double density;             /* hash_bound = floor(density * 2^64), src/read.rs:183      (--density) */
    uint32_t min_abundance;     /* 1..MDBG_MAX_MINABUND                                     (--minabund) */
    uint32_t reads_already_hpc; /* nonzero: skip homopolymer compression                    (--skiphpc) */
    int32_t device;             /* HIP device ordinal, -1 = current device */
    uint32_t flags;             /* MDBG_FLAG_* */
    uint64_t table_capacity_hint; /* expected number of distinct k-min-mers, 0 = size from the data */
    uint32_t scheme;            /* MDBG_SCHEME_*: how minimizers are selected (Read::extract, src/read.rs:85-90) */
    uint32_t syncmer_s;         /* MDBG_SCHEME_SYNCMERS: s-mer length, 0..min(l, 16) with l - s + 1 <= 32 (-s, default 4 in the reference) */
    uint64_t reserved[3];
} mdbg_params;

/* Read-only view of the abundance-filtered node table (what src/main.rs:1014-1117 iterates).
 * Nodes are sorted by `index`.  All arrays are library-owned HOST memory, valid until the next
 * mdbg_finalize / mdbg_reset / mdbg_destroy on the context. */
typedef struct mdbg_nodes {
    uint64_t n;                 /* number of nodes after the abundance filter (main.rs:927) */
    uint32_t k;
    const uint64_t* keys;       /* n*k : canonical k-min-mer (KmerVec::normalize().0) */
    const uint32_t* index;      /* n   : DbgEntry.index = rank of first sighting among ALL distinct k-min-mers (NODE_INDEX) */
    const uint16_t* abundance;  /* n   : DbgEntry.abundance (u16, wraps like the reference) */
    const uint32_t* seqlen;     /* n   : DbgEntry.seqlen of the A-th sighting (main.rs:680-682,778; see n_wrapped) */
    const uint16_t* shift;      /* n*2 : DbgEntry.shift, truncated to u16 (main.rs:675) */
    const uint64_t* shift_full; /* n*2 : un-truncated shift as printed in the .sequences line (main.rs:702) */
    const uint64_t* src_read;   /* n   : ordinal of the read the A-th sighting came from */
    const uint64_t* src_start;  /* n   : raw offset of the node's sequence in that read  (read_offsets.0) */
    const uint64_t* src_end;    /* n   : raw end (exclusive) = pos[i+k-1] + l              (read_offsets.1) */
    const uint8_t* reversed;    /* n   : seq_reversed of that sighting (sequence must be reverse-complemented, main.rs:701) */
    uint64_t n_distinct;        /* "Number of nodes before abundance filter" (main.rs:926) */
    uint64_t n_wrapped;         /* nodes seen >= 65536 times: the abundance wrapped like the reference's u16, and seqlen / shift / src_*
                                 * are those of sighting A + 65536*floor((count-A)/65536): the reference refreshes the entry whenever
                                 * the abundance before the increment equals A-1 (main.rs:676-684) */
} mdbg_nodes;

typedef struct mdbg_stats {
    uint64_t n_reads, n_bases, n_minimizers, n_windows, n_distinct, table_capacity;
    uint64_t n_slow_tiles;      /* tiles that took the generic exact walker (a byte outside ACGT, or > 200 bases of one homopolymer in front of the tile) */
    uint64_t n_tiles;
    double ms_sketch, ms_insert, ms_finalize; /* device time (HIP events) accumulated over calls since create/reset */
    double ms_sketch_tile;      /* of ms_sketch: time inside sketch_tile_kernel launches only (HIP events around each launch) */
    uint64_t n_sketch_tile_launches;
    uint64_t n_sketch_tile_bases; /* raw bases covered by those launches */
    uint64_t tile_bases;        /* raw bases a tile owns under this context's parameters (n_tiles = sum over batches of ceil(batch bases / tile_bases)) */
    uint64_t n_link_matches;    /* only with MDBG_COUNT_LINKS in the environment (test hook; 0 otherwise): repeats of a key confirmed as a LINK of their left neighbour's
                                 * match (one value compared instead of k: csrc/table.hip, upsert_wave), since create / mdbg_reset */
    uint64_t reserved[3];
} mdbg_stats;

mdbg_ctx* mdbg_create(const mdbg_params* p, int* err);
void mdbg_destroy(mdbg_ctx* ctx);

/* Threading: every entry point may be called from any host thread; calls on one context are serialised internally.
 * mdbg_ingest_batch (and mdbg_sketch_only) may be called CONCURRENTLY from several threads, as the reference calls
 * process_read_aux from its --threads workers (src/main.rs:834-913): each caller's PCIe copy runs in its own staging
 * slot and overlaps the kernels of the others.  Batches are ordered by first_read_ordinal, not by call time, so the node
 * table does not depend on the interleaving.  Result buffers (mdbg_sketch_only, mdbg_finalize) belong to the context:
 * callers that need them concurrently use one context each.  mdbg_destroy must not race with other calls.
 *
 * Ingest a batch of reads given as concatenated ASCII (HOST memory) + n_reads+1 offsets.  The bytes are
 * copied to the device and processed there; the sketch of every read stays resident for mdbg_reset. */
int mdbg_ingest_batch(mdbg_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                      uint64_t first_read_ordinal);
/* Same, with both buffers already in DEVICE memory (d_bases 16-byte aligned).  n_bases = offsets[n_reads].
 * offsets[0] may be > 0: bytes in front of the first read are ignored (lets a caller pass an aligned pointer into the
 * middle of a larger buffer). */
int mdbg_ingest_batch_device(mdbg_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads,
                             uint64_t n_bases, uint64_t first_read_ordinal);

/* ---- 2-bit packed input -------------------------------------------------------------------------------------
 * Layout: one uint64_t per 32 bases of the concatenated batch, word w = bases [32w, 32w + 32): bit i (0..31) = bit 1 of the
 * ASCII byte of base 32w + i, bit 32 + i = bit 2 of that byte, i.e. the two bits of the code (ascii >> 1) & 3 (A=0 C=1 T=2
 * G=3) as two 32-bit planes.  Bits past the last base are ignored.  Every byte that is not one of "ACGT" (N, lower case,
 * anything else) is listed in the exception side-list (position in the batch ascending, original byte); its two bits in
 * the words are ignored.  The GPU kernel reads the planes as they are (no unpacking step) and treats tiles that hold an
 * exception with the generic exact walker, so N hashes as 0 and other bytes raise MDBG_E_ALPHABET under the reference's
 * rule, exactly as with ASCII input.  mdbg_pack_reads (mdbg_emit.h, host) and mdbg_pack_device (below) produce the layout. */
typedef struct mdbg_packed_batch {
    const uint64_t* words;      /* (n_bases + 31) / 32 words, 16-byte aligned when in device memory */
    const uint64_t* offsets;    /* n_reads + 1 offsets in BASES into the concatenated batch */
    uint64_t n_reads;
    const uint64_t* exc_pos;    /* n_exc positions, ascending */
    const uint8_t* exc_val;     /* n_exc bytes */
    uint64_t n_exc;
} mdbg_packed_batch;
/* HOST buffers (offsets[0] must be 0); same semantics and thread-safety as mdbg_ingest_batch. */
int mdbg_ingest_batch_packed(mdbg_ctx* ctx, const mdbg_packed_batch* batch, uint64_t first_read_ordinal);
/* DEVICE buffers; n_bases = offsets[n_reads].  _sketch_ runs the sketch stage only (see mdbg_sketch_device). */
int mdbg_ingest_batch_packed_device(mdbg_ctx* ctx, const mdbg_packed_batch* batch, uint64_t n_bases, uint64_t first_read_ordinal);
int mdbg_sketch_packed_device(mdbg_ctx* ctx, const mdbg_packed_batch* batch, uint64_t n_bases, uint64_t first_read_ordinal);
/* ASCII -> packed on the device (DEVICE buffers; d_words holds (n_bases + 31) / 32 words).  Exceptions are appended to
 * d_exc_pos / d_exc_val (room for exc_cap entries, sorted by position on return); *n_exc is their number — if it exceeds
 * exc_cap the lists are incomplete and the call returns MDBG_E_CAPACITY. */
int mdbg_pack_device(mdbg_ctx* ctx, const uint8_t* d_bases, uint64_t n_bases, uint64_t* d_words, uint64_t* d_exc_pos,
                     uint8_t* d_exc_val, uint64_t exc_cap, uint64_t* n_exc);

/* The Read::extract seam alone: sketches a batch (HOST buffers) without touching the node table.
 * Outputs (library-owned host memory, valid until the next call on ctx): hashes[m] = Read.transformed,
 * positions[m] = Read.minimizers_pos, per_read_offsets[n_reads+1] delimiting each read's slice. */
int mdbg_sketch_only(mdbg_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                     const uint64_t** hashes, const uint64_t** positions, const uint64_t** per_read_offsets,
                     uint64_t* n_minimizers);

/* --read_stats (src/main.rs:939-1004, src/read_stats.rs): for every read of the batch (HOST buffers, as mdbg_ingest_batch) and every
 * k-min-mer window of it (reads with MORE than k minimizers only), the abundance of that k-min-mer in the table as it stands after the
 * abundance filter: DbgEntry.abundance (u16) of a node that passes --minabund, else 0.  counts[per_read_offsets[r] ..
 * per_read_offsets[r+1]) are the values of read r in window order - one ".read_stats" line "{id}: c0 c1 ... ".  The batch is sketched
 * on the device and looked up there; neither the table nor the resident sketches change.  Library-owned HOST arrays, valid until the
 * next call on ctx.  Not available on a context that holds routed records. */
int mdbg_query_batch(mdbg_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, const uint32_t** counts,
                     const uint64_t** per_read_offsets, uint64_t* n_windows);

int mdbg_finalize(mdbg_ctx* ctx, mdbg_nodes* out);
/* Same node table, but every pointer in *out is DEVICE memory (no copy to the host). */
int mdbg_finalize_device(mdbg_ctx* ctx, mdbg_nodes* out);
/* The node table for a host that writes the .gfa and nothing else: it stays on the device (where mdbg_graph_edges reads it), and only what the S lines print
 * comes to the host — out->index, out->seqlen, out->abundance (HOST arrays: 10 bytes per node instead of 8 k + 58); every other pointer of *out is NULL.
 * mdbg_emit_write_gfa accepts such a table, the .sequences writer and mdbg_emit_edges do not (they need the minimizer lists: MDBG_E_PARAM). */
int mdbg_finalize_gfa(mdbg_ctx* ctx, mdbg_nodes* out);
/* Order-free digest of a node table whose arrays are in DEVICE memory (mdbg_finalize_device, mdbg_dist_finalize): per node
 *   h = 0x243F6A8885A308D3 ^ abundance;  h = fmix64(h ^ key[j]) for j = 0 .. k-1     (fmix64: the 64-bit finaliser of MurmurHash3)
 * *sum = the sum of h over the nodes mod 2^64, *xr = their XOR.  Equal digests and node counts: the same set of (key, abundance) pairs, i.e. what
 * dbg_nodes holds after the abundance filter (src/main.rs:922-929), whatever the order of the rows; the digests of the ranks' partitions of a multi-GPU
 * run add / XOR up to the digest of the one-GPU table.  Only nodes->n, k, keys, abundance are read.  (bench.py compares it with the CPU oracle's digest
 * of the same reads in every run: a full-size check of the node SET, not of two counts.) */
int mdbg_nodes_digest(mdbg_ctx* ctx, const mdbg_nodes* nodes, uint64_t* sum, uint64_t* xr);
/* Multi-k: keep every cached sketch and all allocations, clear the node table, and re-window the
 * resident sketches with new_k (new_k == 0: also drop the sketches = start over with the same parameters). */
int mdbg_reset(mdbg_ctx* ctx, uint32_t new_k);
/* --lmer-counts (robust minimizers; src/main.rs:544-575, src/minimizers.rs:53-113, src/read.rs:200-205): a minimizer that passed the density
 * threshold is kept only if its l-mer is one of the SELECTED l-mers of a counts file.  codes: the selected l-mers, BOTH orientations
 * listed (the reference inserts an l-mer and its reverse complement), each as a 2-bit code: first base in the highest of the 2*l bits,
 * base code (ascii >> 1) & 3 (A 0, C 1, T 2, G 3); libmdbg_emit's mdbg_lmer_filter_from_counts builds the list from the counts file with
 * the reference's rule.  n = 0 with a non-null pointer = an empty selection (no minimizer survives); codes = NULL switches the filter
 * off.  Density scheme only, l <= 32; call before the first batch (the filter is part of the sketch).  The filter applies to every
 * sketch the context computes afterwards, mdbg_query_batch included. */
int mdbg_set_lmer_filter(mdbg_ctx* ctx, const uint64_t* codes, uint64_t n);
/* Multi-k WITH contig feedback (utils/multik:69-78: every round's input is the original reads plus the previous round's contigs, twice):
 * the reads are ingested once, then mdbg_mark; per round mdbg_rewind(mark) forgets everything ingested after the mark (last round's
 * contigs) and clears the node table, mdbg_reset(k) re-windows what is resident with the round's k, and the round's contigs are
 * ingested as ordinary batches.  The reference reads the contigs BEFORE the reads (they come first in its concatenated file), and
 * DbgEntry.index / the A-th sighting follow the record order: give the reads a first_read_ordinal base above the number of contig
 * records you will ever add (e.g. 1 << 32) and the contigs the ordinals below it — only the order of the ordinals matters.
 * mdbg_rewind(mark) with the same k and no new batches restores exactly the state after the mark once mdbg_reset / mdbg_insert_resident
 * has re-inserted the resident batches.  Not valid while reserved regions are pending (MDBG_E_STATE). */
int mdbg_mark(mdbg_ctx* ctx, uint64_t* mark);
int mdbg_rewind(mdbg_ctx* ctx, uint64_t mark);

int mdbg_get_stats(mdbg_ctx* ctx, mdbg_stats* out);
/* Which HIP events the context records for the ms_* fields of mdbg_stats.  An event is a marker packet with a timestamp on the stream: about 4 microseconds of device
 * time each (scratch/ubench/event_cost.hip), i.e. ~35 microseconds per ingested batch + finalize at level 2 — 1.2 % of a 2.8-ms configs[2] step.
 *   2 (default): around the stages (ms_sketch, ms_insert, ms_finalize) and around every tile-kernel launch (ms_sketch_tile);
 *   1: around the tile-kernel launches only (ms_sketch / ms_insert / ms_finalize stay 0);   0: none (every ms_* field stays 0).
 * Results never depend on it. */
int mdbg_set_timing(mdbg_ctx* ctx, uint32_t level);
const char* mdbg_strerror(int err);
const char* mdbg_last_error(mdbg_ctx* ctx); /* detail of the last failure on ctx ("" if none) */
uint32_t mdbg_abi_version(void);
/* how the library was compiled: bit 0 = the wave-tile kernels of the round-4 experiment are in it (-DMDBG_WAVE_TILES; the default build has ONE tile shape
 * and ignores MDBG_TILE); bit 1: reserved (rounds 4-5 flagged experiment macros with it; the sources carry none since round 6).  The Makefile's build: 0. */
uint32_t mdbg_build_flags(void);
/* Device memory that contexts of this process have released is kept by the library for the next allocation (a hipMalloc that follows large
 * frees takes seconds on this stack); at most MDBG_CACHE_MB megabytes per device (environment; default a third of each device, 0 = keep nothing).
 * This hands all of it back to the runtime and returns the number of bytes released.  Never needed for the correctness of THIS library's calls: one of
 * its allocations that runs out of memory empties the cache itself before it fails.  Another allocator in the same process (torch's caching allocator,
 * RCCL, the host's own hipMalloc) does not know about the cached blocks and can run out of memory against them: such a host calls this after it has
 * destroyed its contexts (or before a large allocation of its own), or caps the cache with MDBG_CACHE_MB. */
uint64_t mdbg_release_cached_memory(void);
/* Host memory for the batches handed to mdbg_ingest_batch / mdbg_ingest_batch_packed: ordinary page-aligned memory that the first ingest call given a pointer
 * into it page-locks (hipHostRegister of the whole allocation, after the caller has written — i.e. faulted in — the batch: 0.011 ms per MB, against 0.24 ms per
 * MB for hipHostMalloc), so that the copy to the device is one DMA (57 GB/s measured) instead of the runtime's staged copy from pageable memory (17 - 20 GB/s
 * for freshly written batches).  A drop-in for malloc / free as far as the caller is concerned; memory from anywhere else is still accepted by the ingest calls
 * and takes the staged path.  mdbg_host_free keeps a
 * page-locked allocation for the next mdbg_host_alloc of about its size (unlocking and unmapping costs what locking did not: 10 ms per 73 MB), up to
 * MDBG_HOST_CACHE_MB megabytes per process (environment; default 2048, 0 = give everything back at once); mdbg_release_cached_memory releases these too.  mdbg_reader_set_allocator (mdbg_emit.h) takes this pair.  mdbg_host_is_pinned: 1 once the
 * allocation that holds p has been page-locked. */
void* mdbg_host_alloc(size_t bytes);
void mdbg_host_free(void* p);
int mdbg_host_is_pinned(const void* p);

/* ---- device-resident stage entry points (used by the multi-GPU driver and the benchmark) -------------
 * Sketch stage only, device buffers in, results appended to the context's resident sketch store. */
int mdbg_sketch_device(mdbg_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads,
                       uint64_t n_bases, uint64_t first_read_ordinal);
/* Window + insert every sketch appended since the last call of this function / reset. */
int mdbg_insert_resident(mdbg_ctx* ctx);
/* ---- multi-GPU: one process per GPU, reads sharded by record, ONE all-to-all of k-min-mer records ----------
 * (the exchanges themselves are done by the host driver over RCCL, see rust_mdbg_amd/dist.py)
 *
 * Key-range routing: packs every k-min-mer occurrence of the sketches not yet inserted into `world` (<= 64)
 * destination buckets, owner = mulhi64(keyhash, world).  One record is k+2 u64: canonical key, the global ordinal
 * (read ordinal << 26 | window index), the key hash.  *d_records receives a DEVICE pointer to the bucketed records
 * (library-owned, valid until the next route_pack/reset), counts[world] (HOST) the records per destination. */
int mdbg_route_pack(mdbg_ctx* ctx, uint32_t world, const uint64_t** d_records, uint64_t* counts);
/* Owner side: room for n more records at the tail of the context's arena (DEVICE pointer, valid until the next
 * arena_reserve / insert_records / reset): receive the all-to-all straight into it, then pass the same pointer to
 * mdbg_insert_records and no copy is made. */
int mdbg_arena_reserve(mdbg_ctx* ctx, uint64_t n_records, uint64_t** d_tail);
/* Owner side: insert records received from peers (DEVICE memory, k+2 u64 each).  Unless they already sit in the arena
 * tail (mdbg_arena_reserve) they are copied into it, so the caller's buffer may be released when the call returns.  After the first call the
 * context is in routed mode: mdbg_finalize is replaced by the four calls below. */
int mdbg_insert_records(mdbg_ctx* ctx, const uint64_t* d_records, uint64_t n_records);
/* Owner side: the table's entries as two query lists, each bucketed by the rank that has to answer (DEVICE pointers,
 * library-owned until the next export / finalize / reset).  Ranks are described by spans of read ordinals sorted by
 * start: span i = [span_lo[i], span_lo[i+1]) belongs to rank span_rank[i].
 *   list A, one entry per distinct k-min-mer, bucketed by the rank of its FIRST sighting: d_first[n_all] (ordinal),
 *           d_solid[n_all] (1 = passes the abundance filter, src/main.rs:922-929); counts_all[r] entries for rank r.
 *   list S, one entry per solid k-min-mer, bucketed by the rank of its A-th sighting: d_ath[n_solid] (ordinal),
 *           d_count[n_solid] (occurrences), d_slot[n_solid] (handle for mdbg_routed_keys), d_idx_all[n_solid] (position of
 *           the same k-min-mer in list A); counts_solid[r] entries for rank r. */
typedef struct mdbg_routed_lists {
    uint64_t n_all, n_solid;
    const uint64_t* d_first; const uint8_t* d_solid;
    const uint64_t* d_ath; const uint32_t* d_count; const uint64_t* d_slot; const uint64_t* d_idx_all;
    uint64_t counts_all[64], counts_solid[64];
} mdbg_routed_lists;
int mdbg_routed_export(mdbg_ctx* ctx, uint32_t world, const uint64_t* span_lo, const uint32_t* span_rank, uint32_t n_spans,
                       mdbg_routed_lists* out);
/* Generator side (the rank whose reads the ordinals belong to): for n first-sighting ordinals (DEVICE) with their solid
 * flags, rank_first[i] = number of queried ordinals smaller than ord[i] (-> DbgEntry.index once the totals of the ranks
 * holding earlier reads are added), rank_solid[i] = the same among solid ones (-> row of the node in index order).
 * ALL queries for this rank must be passed in one call.  Outputs are caller-owned DEVICE buffers of n u64. */
int mdbg_resolve_first(mdbg_ctx* ctx, const uint64_t* d_ord, const uint8_t* d_solid, uint64_t n, uint64_t* d_rank_first,
                       uint64_t* d_rank_solid, uint64_t* total_first, uint64_t* total_solid);
/* Generator side: what the reference stores from the sighting with ordinal ord[i] (src/main.rs:675-684,778), 6 u64 per
 * query into the caller-owned DEVICE buffer d_meta: {seqlen | reversed << 32, shift0, shift1, src_read, src_start,
 * src_end}; all zero for ord[i] == ~0. */
int mdbg_resolve_meta(mdbg_ctx* ctx, const uint64_t* d_ord, uint64_t n, uint64_t* d_meta);
/* Owner side: canonical keys (k u64 each) of the given slot handles into the caller-owned DEVICE buffer d_keys. */
int mdbg_routed_keys(mdbg_ctx* ctx, const uint64_t* d_slot, uint64_t n, uint64_t* d_keys);
/* ---- graph edges (replaces the single-threaded edge loop of src/main.rs:1017-1117) ------------------------------
 * Edges of the node table produced by the LAST mdbg_finalize / mdbg_finalize_device on this context (the table stays on
 * the device): for every node n1 and both of its (k-1)-mers, the nodes listing the same normalized (k-1)-mer, the four
 * orientation tests (main.rs:1062-1075), abundance presimplification (main.rs:1078-1090 and 1104-1115; presimp = 0
 * disables it, the reference's default is 0.01) and overlap = min(n1.seqlen - shift(ori1), n2.seqlen - 1).  n1/n2 are
 * DbgEntry.index values, o1/o2 are '+' / '-'.  The order is the one a sequential pass over the node table produces
 * (n1 in table order, suffix key before prefix key, listings in table order, orientations ++, +-, -+, --); the
 * reference iterates a DashMap, so only the multiset of its L lines is defined.
 * mdbg_graph_edges: HOST arrays owned by the context; mdbg_graph_edges_device: DEVICE arrays; both valid until the next
 * edge call.  The layout equals mdbg_edges of include/mdbg_emit.h, so the host copy can go straight to mdbg_emit_write_gfa. */
typedef struct mdbg_edge_list {
    uint64_t n;
    const uint32_t* n1; const uint8_t* o1; const uint32_t* n2; const uint8_t* o2; const uint32_t* overlap;
    uint64_t presimp_removed;     /* candidate edges dropped by the abundance rule (before the symmetric removal) */
} mdbg_edge_list;
int mdbg_graph_edges(mdbg_ctx* ctx, float presimp, mdbg_edge_list* out);
int mdbg_graph_edges_device(mdbg_ctx* ctx, float presimp, mdbg_edge_list* out);

/* ---- multi-GPU, second mode: replicated sketches, partitioned table ----------------------------------------
 * Within one node the sketch is much more compact than the k-min-mers cut from it (every minimizer sits in k windows),
 * so the ranks may exchange SKETCHES instead (one all-gather), each rank then windows the global sketch but inserts only
 * the k-min-mers it owns.  Ownership is an O(1) function of the canonical key (its two ends and its middle), the table,
 * insertion and finalize are the single-GPU ones, and nothing is approximate.  DbgEntry.index needs the first sightings
 * of all ranks: mdbg_finalize_begin exposes this rank's first-sighting / solid bitmaps over the (identical on every
 * rank) global sketch; the driver sums them across ranks (bits are disjoint) and mdbg_finalize_end emits the rows. */
int mdbg_set_partition(mdbg_ctx* ctx, uint32_t world, uint32_t rank);   /* before the first insertion */
typedef struct mdbg_sketch_store {
    uint64_t n_minimizers, n_reads;
    const uint64_t* d_hashes;        /* [n_minimizers] DEVICE */
    const uint32_t* d_positions;     /* [n_minimizers] raw position relative to the read */
    const uint64_t* d_read_offsets;  /* [n_reads + 1] first minimizer of every read */
} mdbg_sketch_store;
int mdbg_sketch_view(mdbg_ctx* ctx, mdbg_sketch_store* out);  /* the resident store (valid until the next sketch/ingest/reset) */
/* Append an already computed sketch (DEVICE arrays, e.g. another rank's mdbg_sketch_view): n_reads reads whose minimizers
 * are hashes/positions[read_offsets[r] .. read_offsets[r+1]).  Counterpart of mdbg_sketch_device without the kernel. */
int mdbg_ingest_sketch(mdbg_ctx* ctx, const uint64_t* d_hashes, const uint32_t* d_positions, const uint64_t* d_read_offsets,
                       uint64_t n_reads, uint64_t first_read_ordinal);
/* Zero-copy variant for receives that can write straight into the resident store (RCCL recv, peer copies):
 *   mdbg_store_reserve   sizes the store up front (total minimizers / reads it will hold), so that it never has to move;
 *   mdbg_sketch_reserve  appends an UNCOMMITTED region of n_minimizers entries and returns where to write its hashes and
 *                        positions (DEVICE pointers; they stay valid because the store refuses to grow, MDBG_E_STATE, while
 *                        regions are pending); *region identifies it;
 *   mdbg_sketch_commit   registers [region, region + n_minimizers) - or any part of a reserved region - as the sketch of
 *                        n_reads reads; d_read_offsets are relative to `region` ([0] = 0, [n_reads] = n_minimizers, checked
 *                        on the device: a violation is reported as MDBG_E_PARAM by the next mdbg_insert_resident).  The
 *                        call is stream-ordered: the offsets buffer must stay alive until the next synchronising call.
 *                        owned_windows: see mdbg_owner_counts.
 *   mdbg_last_batch      describes the batch registered last (e.g. the one mdbg_sketch_device just produced), with DEVICE
 *                        pointers to its part of the store: what a rank sends to its peers.  Synchronises the context's
 *                        stream, so the arrays may be read from any other stream afterwards. */
int mdbg_store_reserve(mdbg_ctx* ctx, uint64_t n_minimizers_total, uint64_t n_reads_total);
int mdbg_sketch_reserve(mdbg_ctx* ctx, uint64_t n_minimizers, uint64_t** d_hashes, uint32_t** d_positions, uint64_t* region);
int mdbg_sketch_commit(mdbg_ctx* ctx, uint64_t region, uint64_t n_minimizers, const uint64_t* d_read_offsets, uint64_t n_reads,
                       uint64_t first_read_ordinal, uint64_t owned_windows);
/* Windows of the batch registered last, counted per owning rank (counts[world], HOST): computed once by the rank that
 * sketched the batch and shipped with it (mdbg_sketch_commit's owned_windows = counts[receiver]; UINT64_MAX = unknown), so
 * that the receivers size their tables without re-counting a foreign sketch.  When world equals this context's partition,
 * the call also records the context's own share for its own batch.  The counts are verified against what is actually
 * inserted (MDBG_E_PARAM from mdbg_insert_resident on a mismatch). */
int mdbg_owner_counts(mdbg_ctx* ctx, uint32_t world, uint64_t* counts);
/* Owner LISTS: like mdbg_owner_counts, and in addition the windows themselves, bucketed by owning rank: d_lists (DEVICE, library-owned
 * until the next call of this function) holds counts[0] entries for rank 0, then counts[1] for rank 1, ...; an ENTRY is a pair of
 * uint32_t: the index of the window's first minimizer and the index of its read, both relative to the batch (so rank r's bucket starts
 * at d_lists + 2 * (counts[0] + ... + counts[r-1]) and is 8 * counts[r] bytes).  The sender ships bucket r to rank r with the sketch
 * (8 bytes per window) and the receiver registers it with mdbg_sketch_commit_listed (n_list = number of entries): it then inserts
 * exactly its windows — no scan of the foreign sketch, no minimizer -> read map built for it — so the per-rank work does not grow with
 * the number of ranks.  world <= 64.  Buckets are ordered by spans of 2048 window starts (any order inside a span); every entry is
 * verified at insertion (range, read, ownership, and the total count: MDBG_E_PARAM from mdbg_insert_resident on a mismatch).  The list
 * is copied by mdbg_sketch_commit_listed: the caller's buffer may be reused once the context's stream has passed the call (mdbg_sync). */
int mdbg_owner_lists(mdbg_ctx* ctx, uint32_t world, uint64_t* counts, const uint32_t** d_lists);
int mdbg_sketch_commit_listed(mdbg_ctx* ctx, uint64_t region, uint64_t n_minimizers, const uint64_t* d_read_offsets, uint64_t n_reads,
                              uint64_t first_read_ordinal, const uint32_t* d_list, uint64_t n_list);
typedef struct mdbg_batch_info {
    uint64_t store_offset, n_minimizers, first_slot, n_reads, first_read_ordinal;
    const uint64_t* d_hashes;        /* [n_minimizers] */
    const uint32_t* d_positions;     /* [n_minimizers] */
    const uint64_t* d_read_offsets;  /* [n_reads + 1], absolute: subtract store_offset for offsets relative to the batch */
} mdbg_batch_info;
int mdbg_last_batch(mdbg_ctx* ctx, mdbg_batch_info* out);
int mdbg_finalize_begin(mdbg_ctx* ctx, uint64_t** d_bm_first, uint64_t** d_bm_solid, uint64_t* n_words);
/* out: DEVICE pointers, this rank's nodes in table order; d_row[i] = global row (position in index order);
 * out->n_distinct and *n_nodes_global are the totals over all ranks (from the merged bitmaps). */
int mdbg_finalize_end(mdbg_ctx* ctx, mdbg_nodes* out, const uint64_t** d_row, uint64_t* n_nodes_global);

/* Wait for all device work queued by ctx. */
int mdbg_sync(mdbg_ctx* ctx);
/* Plain copies between host memory and device buffers handed out by / given to this library. */
int mdbg_copy_to_host(mdbg_ctx* ctx, void* dst, const void* d_src, uint64_t nbytes);
int mdbg_copy_to_device(mdbg_ctx* ctx, void* d_dst, const void* src, uint64_t nbytes);

/* Synthetic HiFi-shaped reads (counter-based, integer-only generator; the same bytes can be regenerated
 * on the CPU, see rust_mdbg_amd/synth.py).  Fills library-owned DEVICE buffers. */
typedef struct mdbg_synth_params {
    uint64_t seed;
    uint64_t genome_len;
    uint64_t n_reads;
    uint32_t mean_len, sd_len, min_len, max_len;
    uint32_t err_ppm;           /* per-base error rate in parts per million, split sub/ins/del 1:1:1 */
    uint32_t reserved;
} mdbg_synth_params;
int mdbg_synth_reads_device(mdbg_ctx* ctx, const mdbg_synth_params* sp, uint64_t first_read,
                            const uint8_t** d_bases, const uint64_t** d_offsets, uint64_t* n_bases);

/* Measurement hook (scratch/measure_rank_w8.py; no counterpart in the reference): milliseconds the multi-GPU layer's sender (out[0]) and receiver (out[1]) spend on the
 * segments of the batch registered last for `world` ranks; counts / d_lists as mdbg_owner_lists returned them, `skip` = the bucket that ships nothing;
 * out[2] = list entries, out[3] = hashes packed. */
int mdbg_dbg_segments_ms(mdbg_ctx* ctx, uint32_t world, uint32_t skip, const uint64_t* counts, const uint32_t* d_lists, double* out);

#ifdef __cplusplus
}
#endif
#endif /* MDBG_HIP_H */
