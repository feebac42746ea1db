/* mdbg_dist.h — multi-GPU layer of libmdbg_hip.so behind a plain C ABI: one process (or thread) per GPU, reads sharded by record,
 * the k-min-mer table partitioned by key, exchanges done by the library itself over RCCL.
 *
 * What it replaces: rust-mdbg has no multi-process path (one process, --threads workers, one DashMap: src/main.rs:834); the
 * partitioning is the north star's: every rank sketches its own reads (mdbg_hip.h), the ranks exchange what the owners of the
 * k-min-mers need, every rank counts the keys it owns, and DbgEntry.index (NODE_INDEX, src/main.rs:598,661) is made global by
 * summing the ranks' first-sighting bitmaps.  Mode implemented here: the SKETCH exchange (profiles/r02_notes.md: 3x faster per rank
 * than routing expanded k-min-mer records): per round every rank sends each peer the list of the windows that peer owns (8 bytes per window,
 * mdbg_owner_lists; a k-min-mer belongs to the rank its smallest minimizer hash maps to) and the HASHES those windows need — by default only
 * those (segments: a peer's windows come in runs, a run of r windows needs r + k - 1 hashes), see mdbg_dist_set_exchange.  The raw
 * positions stay home: a node's seqlen / shift / origin need four positions of ONE window, which mdbg_dist_finalize fetches from the
 * rank that sketched the read — a few MB per finalize instead of 4 bytes per minimizer per round.  All of a round travels
 * in ONE grouped set of ncclSend / ncclRecv pairs —
 * xGMI is point to point, every pair of GPUs uses its own link — and the receives land directly in reserved regions of the
 * resident sketch store (mdbg_sketch_reserve: no staging copy).  Results are identical to a single context fed all reads
 * (tests/test_gpu_dist_c.py, examples/mdbg_dist_threads.c).
 *
 * Every call below is COLLECTIVE: all ranks call it the same number of times, in the same order (a rank that has run out of reads
 * passes n_reads = 0).  Error codes and conventions are those of mdbg_hip.h.
 * The first round compares library version (MDBG_ABI_VERSION), exchange mode, chunks per call and the sketch parameters over the ranks: a
 * rank that differs makes every rank's call return MDBG_E_PARAM.  The same round measures where the windows' smallest hashes fall and deals the
 * value range out to the ranks evenly (mdbg_dist_set_exchange); that assignment holds for the lifetime of the mdbg_dist.
 */
#ifndef MDBG_DIST_H
#define MDBG_DIST_H

#include <stdint.h>

#include "mdbg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one transfer of a grouped exchange: `bytes` bytes of DEVICE memory to / from rank `peer` */
typedef struct mdbg_xfer { uint32_t peer; void* d_ptr; uint64_t bytes; } mdbg_xfer;

/* Communicator.  mdbg_comm_rccl fills it with direct RCCL calls; a host with another transport (or a test that runs several ranks
 * as threads of one process on one GPU) supplies its own three functions.  All functions return 0 or a negative MDBG_E_* code and
 * block the calling host thread until the data has landed. */
typedef struct mdbg_comm {
    void* self;
    uint32_t rank, world;
    /* every rank contributes n values (HOST memory); recv (HOST, world * n values) receives them ordered by rank */
    int (*allgather_u64)(void* self, const uint64_t* send, uint32_t n, uint64_t* recv);
    /* all transfers of one round at once; transfers between a pair of ranks are matched in the order they are listed; zero-byte
     * transfers are never listed */
    int (*exchange)(void* self, const mdbg_xfer* sends, uint32_t n_sends, const mdbg_xfer* recvs, uint32_t n_recvs);
    /* in-place element-wise sum over the ranks of n u64 values in DEVICE memory */
    int (*allreduce_sum_u64)(void* self, uint64_t* d_buf, uint64_t n);
    /* optional (both or none; NULL = not available): the same exchange split in two, so that the library can keep its GPU busy while
     * the data travels — exchange_begin starts all transfers and returns, exchange_wait blocks until every one of them has landed.
     * At most one exchange is in flight per communicator. */
    int (*exchange_begin)(void* self, const mdbg_xfer* sends, uint32_t n_sends, const mdbg_xfer* recvs, uint32_t n_recvs);
    int (*exchange_wait)(void* self);
} mdbg_comm;

/* RCCL transport: nccl_comm is an initialised ncclComm_t of `world` ranks whose rank `rank` is this process' GPU (rccl.h:220
 * ncclCommInitRank).  The library resolves ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd / ncclAllGather / ncclAllReduce from the
 * RCCL already loaded into the process (or librccl.so.1) at run time, so libmdbg_hip.so itself does not depend on RCCL.
 * The ncclComm_t stays the caller's (destroy it after mdbg_dist_destroy); the transport object behind out->self (a stream and two
 * small staging buffers) passes to the mdbg_dist created with it and is freed by mdbg_dist_destroy: one mdbg_comm_rccl call per mdbg_dist. */
int mdbg_comm_rccl(void* nccl_comm, uint32_t rank, uint32_t world, mdbg_comm* out);

typedef struct mdbg_dist mdbg_dist;

/* p as for mdbg_create (p->device = this rank's GPU).  The comm table is copied. */
mdbg_dist* mdbg_dist_create(const mdbg_params* p, const mdbg_comm* comm, int* err);
void mdbg_dist_destroy(mdbg_dist* d);
/* the rank's local context: for mdbg_get_stats, mdbg_sync, mdbg_last_error, mdbg_synth_reads_device ... (do not ingest through it;
 * mdbg_query_batch on it returns MDBG_E_STATE: the table holds only the keys this rank owns) */
mdbg_ctx* mdbg_dist_ctx(mdbg_dist* d);

/* Pipelining inside one ingest call: the batch is cut into `chunks` runs of whole reads (every rank runs `chunks` rounds per call, so all
 * ranks must set the same value); while chunk i travels to the peers (exchange_begin ... exchange_wait of the communicator) the tile
 * kernel already works on chunk i+1, and the windows are inserted when the last chunk has arrived.  xGMI moves a rank's share of a
 * round in about the time the sketch kernel needs for it, so this hides most of the exchange.  Default 1 (no cutting); 1..64. */
int mdbg_dist_set_pipeline(mdbg_dist* d, uint32_t chunks);

/* What a round ships to the peers (the same value on every rank; not while a round is in flight):
 * MDBG_EXCHANGE_SEGMENTS (default): per peer the list of the windows it owns (8 bytes each) and ONLY the hashes those windows need.  A
 *   k-min-mer is owned by the rank its smallest minimizer hash maps to; consecutive windows of a read share that hash for about (k + 1) / 2
 *   steps, so a rank's windows come in runs and a run of r windows needs r + k - 1 hashes: a few hashes per window, and a volume per rank
 *   that does not grow with the number of ranks.  The foreign sketches are then resident only where this k needs them: mdbg_dist_reset(d, new_k != k)
 *   runs the rounds' exchange again for the new k (the reads are not sketched again; neither k's hashes are a subset of the other's — the owner of a window is
 *   a function of the smallest of ITS k hashes).
 * MDBG_EXCHANGE_WHOLE: every rank receives every sketch entire (8 bytes per minimizer and peer: what round 2 did).  THE MODE OF A MULTI-K SWEEP
 *   (utils/multik: k = 10, 15, .. 40 on the same reads): the resident global sketch is re-windowed at every k without a new exchange — one
 *   exchange of 1.3 GB into a rank (8 ranks, 7-Gbase shards) instead of one of 0.3 GB per k under segments (seven k: 2.1 GB).  bench.py --multik selects it
 *   (--multik-exchange segments: the other way). */
enum { MDBG_EXCHANGE_SEGMENTS = 0, MDBG_EXCHANGE_WHOLE = 1 };
int mdbg_dist_set_exchange(mdbg_dist* d, uint32_t mode);

/* One round: sketch this rank's batch (DEVICE buffers, as mdbg_ingest_batch_device / mdbg_ingest_batch_packed_device), exchange
 * sketches and window lists with every peer, insert the windows this rank owns.  first_read_ordinal is GLOBAL (position of the
 * batch's first record in the whole input), ordinal ranges of different ranks and rounds must not overlap. */
int mdbg_dist_ingest_batch_device(mdbg_dist* d, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases,
                                  uint64_t first_read_ordinal);
int mdbg_dist_ingest_batch_packed_device(mdbg_dist* d, const mdbg_packed_batch* batch, uint64_t n_bases, uint64_t first_read_ordinal);

/* This rank's partition of the node table (DEVICE arrays as mdbg_finalize_device, in table order): d_row[i] = position of node i in
 * the global table (= rank in DbgEntry.index order), so concatenating the partitions of all ranks and ordering by d_row gives the
 * single-GPU table; out->index holds the GLOBAL DbgEntry.index; out->n_distinct and *n_nodes_global are totals over all ranks. */
int mdbg_dist_finalize(mdbg_dist* d, mdbg_nodes* out, const uint64_t** d_row, uint64_t* n_nodes_global);
/* new_k = 0: drop everything; else the node table of the same reads at another k.  COLLECTIVE.  MDBG_EXCHANGE_WHOLE: the resident GLOBAL sketch is re-windowed
 * (no exchange: every rank holds every hash).  MDBG_EXCHANGE_SEGMENTS: every round is exchanged again for the new k from the sketches the ranks hold of their
 * own reads — new window lists, new segments into the same regions of the store.  Either way nothing is sketched again, and the positions the new nodes need
 * are fetched at the next finalize. */
int mdbg_dist_reset(mdbg_dist* d, uint32_t new_k);

/* Bytes this rank has received / sent through the communicator's `exchange` since mdbg_dist_create or the last mdbg_dist_reset(d, 0)
 * (sketch rounds + position fetches), and the number of position queries it sent: what a link budget is made of.  Any pointer may be NULL. */
int mdbg_dist_traffic(mdbg_dist* d, uint64_t* bytes_in, uint64_t* bytes_out, uint64_t* position_queries);

/* Host milliseconds this rank spent per stage of its rounds (sketch, wait for the exchange, scatter, commit, owner lists, pack, size all-gather, reserve, sync, exchange
 * begin, insertion) and of its finalize calls (begin, bitmap all-reduce, end, of which position fetch), summed since create or the last call with reset != 0; out[i] belongs to
 * mdbg_dist_stage_name(i) (null from the first index that has no stage).  No counterpart in the reference (one process, src/main.rs:834): bench.py prints these beside the
 * eight-rank budget of DESIGN.md 3.4 so that a multi-GPU result can be attributed to a stage. */
int mdbg_dist_stage_ms(mdbg_dist* d, double* out, uint32_t n, uint64_t* n_rounds, int reset);
const char* mdbg_dist_stage_name(uint32_t i);

#ifdef __cplusplus
}
#endif
#endif /* MDBG_DIST_H */
