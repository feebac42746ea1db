"""End-to-end host pipeline above the two C ABIs: reads file -> GPU hot path -> .gfa + .sequences.

The same stages as rust-mdbg's main() for the default (density) scheme: parse (a reader thread runs ahead of the
GPU like seq_io's reader thread, src/main.rs:830-839), process_read_aux on batches, abundance filter, graph emit
(src/main.rs:1006-1117), and the .sequences file (one LZ4-frame file instead of one per worker thread)."""
import queue
import threading
import time

from .api import Mdbg
from .emit import Emitter, Reader, lmer_filter_from_counts


def apply_lmer_counts(m, lmer_counts, l, density, lmer_counts_min, lmer_counts_max):
    """--lmer-counts FILE [--lmer_counts_min A --lmer_counts_max B] (src/main.rs:392-409, 499-503): restrict the sketch of context m"""
    if lmer_counts is None:
        return
    codes, _ = lmer_filter_from_counts(lmer_counts, l, density, lmer_counts_min, lmer_counts_max)
    m.set_lmer_filter(codes)


def run_file(path, prefix, k, l, density, min_abundance=2, reads_already_hpc=False, presimp=0.01, batch_bases=256 << 20,
             strip_newlines=False, device=-1, write_sequences=True, lmer_counts=None, lmer_counts_min=2, lmer_counts_max=100000,
             threads=1, packed=None):
    """-> dict of counters (what the reference prints: reads, nodes before/after filter, edges, presimp removals).
    threads: host threads of the reader (uncompressed input: mdbg_reader_open_mt) and of the 2-bit packer; packed: hand the GPU 2-bit
    packed batches (a quarter of the bytes over PCIe), default: when threads > 1"""
    if packed is None:
        packed = threads > 1
    q = queue.Queue(maxsize=2)
    stop = threading.Event()               # set when the consumer gives up: the reader must not stay blocked in put()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    free = threading.Semaphore(2)          # packed batches are views into the reader's two alternating buffer sets: at most two may be outstanding

    def produce():
        try:
            with Reader(path, strip_newlines, threads=threads, device_buffers=True) as r:
                if packed:
                    # the reader packs while it parses (mdbg_reader_next_packed); batch i+1 is produced while batch i is ingested
                    it = r.batches_packed(batch_bases, copy=False)
                    while True:
                        while not free.acquire(timeout=0.2):
                            if stop.is_set():
                                return
                        pk = next(it, None)
                        if pk is None:
                            break
                        if not put((pk, None, pk["n_bases"])):
                            return
                    put(None)
                    stop.wait()                   # the reader's buffers must outlive the last batch's ingest: wait until the consumer is done
                    return
                if r.parallel:
                    # ASCII batches of the parallel reader are views into its two alternating buffers: no copy, at most two outstanding (as the packed ones);
                    # until round 5 every batch was copied once more on this thread (7 GB of memcpy per 7 Gbases: most of the ASCII path's time)
                    it = r.batches(batch_bases, copy=False)
                    while True:
                        while not free.acquire(timeout=0.2):
                            if stop.is_set():
                                return
                        item = next(it, None)
                        if item is None:
                            break
                        if not put((item[0], item[1].copy(), len(item[0]))):
                            return
                    put(None)
                    stop.wait()
                    return
                for bases, offs in r.batches(batch_bases):
                    if not put((bases, offs, len(bases))):
                        return
            put(None)
        except BaseException as e:          # noqa: BLE001
            put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    n_reads = n_bases = 0
    tm = {}
    t0 = time.perf_counter()
    try:
        with Mdbg(k, l, density, min_abundance, reads_already_hpc=reads_already_hpc, device=device) as m:
            apply_lmer_counts(m, lmer_counts, l, density, lmer_counts_min, lmer_counts_max)
            tm["open"] = time.perf_counter() - t0
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                payload, offs, nb = item
                if packed:
                    m.ingest_packed(payload, n_reads)
                    free.release()                      # this batch's buffers may be reused
                else:
                    m.ingest(payload, offs, n_reads)    # ctypes releases the GIL: the reader thread parses the next batch meanwhile
                    free.release()
                n_reads += (len(payload["offsets"]) if packed else len(offs)) - 1
                n_bases += nb
            tm["ingest"] = time.perf_counter() - t0
            # without the .sequences pass the host needs three columns of the node table (the S lines); the minimizer lists stay on the device, where the edge
            # stage reads them (the full copy-out was 150 MB for 465 k nodes: 40 ms of a 190-ms run)
            nodes = m.finalize(gfa_only=not write_sequences)
            stats = m.stats()
            tm["finalize"] = time.perf_counter() - t0
            # edges on the GPU from the device-resident node table (the reference's single-threaded loop, src/main.rs:1017-1117);
            # the host copy of the list goes straight into the GFA writer
            raw = m.graph_edges(presimp, raw=True)
            edges = dict(n1=[0] * int(raw.n), presimp_removed=int(raw.presimp_removed))
            tm["edges"] = time.perf_counter() - t0
            em = Emitter()
            em.write_gfa(prefix + ".gfa", nodes, raw)
            tm["gfa"] = time.perf_counter() - t0
        tm["close"] = time.perf_counter() - t0
    finally:
        stop.set()                          # error or not: release the reader (it closes the file) and wait for it
        th.join()
    tm["reader_closed"] = time.perf_counter() - t0
    if write_sequences:                              # second pass over the input for the node sequences
        def again():
            first = 0
            with Reader(path, strip_newlines, threads=threads) as r:
                for bases, offs in r.batches(batch_bases, copy=False):      # consumed before the next batch is asked for
                    yield bases, offs, first
                    first += len(offs) - 1
        t1 = time.perf_counter()
        if threads > 1:                              # one file per writer thread, like the reference's worker threads (main.rs:614-630)
            em.write_sequences_parallel(prefix, nodes, l, again(), min(threads, 16))
        else:
            em.write_sequences(prefix + ".0.sequences", nodes, l, again())
        tm["sequences"] = time.perf_counter() - t1
    return dict(n_reads=n_reads, n_bases=n_bases, n_minimizers=stats["n_minimizers"], n_windows=stats["n_windows"],
                n_nodes_before=nodes["n_nodes_before"], n_nodes=nodes["n_nodes"], n_edges=len(edges["n1"]),
                presimp_removed=edges["presimp_removed"], seconds_until={k_: round(v, 4) for k_, v in tm.items()})


READ_ORDINAL_BASE = 1 << 32     # contig feedback: the reads' ordinals start here, the contigs of a round take [0, 2 * n_contigs)


def concat_records(seqs):
    """list of bytes -> (uint8 bases, uint64 offsets) as the ingest calls take them"""
    import numpy as np
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(x) for x in seqs], dtype=np.uint64)
    return np.frombuffer(b"".join(seqs), dtype=np.uint8), offs


def run_multik(path, prefix, ks, l, density, min_abundance=2, reads_already_hpc=False, presimp=0.01, batch_bases=256 << 20,
               strip_newlines=False, device=-1, contigs_fn=None, min_contig_len=100000, lmer_counts=None, lmer_counts_min=2, lmer_counts_max=100000):
    """One pass over the reads, one graph per k (the k sweep of the reference's utils/multik:69-78): the reads are sketched once,
    the sketches stay resident on the GPU, and every k only clears and refills the counting table (mdbg_reset) and rebuilds nodes
    and edges.  Writes <prefix>-k<k>.gfa; -> {k: counters}.

    contigs_fn(k, gfa_path, nodes) -> list of bytes: the script's contig feedback.  The simplification that turns a round's graph
    into contigs is the caller's (the script shells out to magic_simplify = gfatools + to_basespace, outside this path); what it
    returns is filtered like `seqtk seq -L 100000` (min_contig_len), taken TWICE (`zcat -f x.msimpl.fa x.msimpl.fa`, utils/multik:72)
    and put IN FRONT of the reads for the next k: the reads keep their resident sketches and their ordinals (READ_ORDINAL_BASE + i),
    the contigs get the ordinals 0 .. 2C-1, and the previous round's contigs are forgotten (mdbg_rewind)."""
    ks = list(ks)
    out = {}
    n_reads = n_bases = 0
    base = READ_ORDINAL_BASE if contigs_fn is not None else 0
    with Mdbg(ks[0], l, density, min_abundance, reads_already_hpc=reads_already_hpc, device=device) as m, Reader(path, strip_newlines) as r:
        apply_lmer_counts(m, lmer_counts, l, density, lmer_counts_min, lmer_counts_max)
        for bases, offs in r.batches(batch_bases):
            m.ingest(bases, offs, base + n_reads)
            n_reads += len(offs) - 1
            n_bases += len(bases)
        mark = m.mark()
        em = Emitter()
        contigs = []
        for i, k in enumerate(ks):
            if i:
                if contigs_fn is not None:
                    m.rewind(mark)                       # last round's contigs go, the reads' sketches stay
                m.reset(k)                               # sketches stay; windows of the new k are inserted again
                if contigs:
                    twice = contigs + contigs
                    cb, co = concat_records(twice)
                    m.ingest(cb, co, 0)
            nodes = m.finalize()
            raw = m.graph_edges(presimp, raw=True)
            gfa = "%s-k%d.gfa" % (prefix, k)
            em.write_gfa(gfa, nodes, raw)
            st = m.stats()
            out[k] = dict(n_reads=n_reads, n_bases=n_bases, n_contigs=len(contigs), n_minimizers=st["n_minimizers"], n_windows=st["n_windows"],
                          n_nodes_before=nodes["n_nodes_before"], n_nodes=nodes["n_nodes"], n_edges=int(raw.n),
                          presimp_removed=int(raw.presimp_removed))
            if contigs_fn is not None:
                contigs = [bytes(c) for c in contigs_fn(k, gfa, nodes) if len(c) >= min_contig_len]
    return out
