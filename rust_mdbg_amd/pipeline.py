"""End-to-end host pipeline above the two C ABIs: reads file -> GPU hot path -> .gfa + .sequences.

The same stages as rust-mdbg's main() for the default (density) scheme: parse (a reader thread runs ahead of the
GPU like seq_io's reader thread, src/main.rs:830-839), process_read_aux on batches, abundance filter, graph emit
(src/main.rs:1006-1117), and the .sequences file (one LZ4-frame file instead of one per worker thread)."""
import queue
import threading

from .api import Mdbg
from .emit import Emitter, Reader


def run_file(path, prefix, k, l, density, min_abundance=2, reads_already_hpc=False, presimp=0.01, batch_bases=256 << 20,
             strip_newlines=False, device=-1, write_sequences=True):
    """-> dict of counters (what the reference prints: reads, nodes before/after filter, edges, presimp removals)"""
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

    def produce():
        try:
            with Reader(path, strip_newlines) as r:
                for item in r.batches(batch_bases):
                    if not put(item):
                        return
            put(None)
        except BaseException as e:          # noqa: BLE001
            put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    n_reads = n_bases = 0
    try:
        with Mdbg(k, l, density, min_abundance, reads_already_hpc=reads_already_hpc, device=device) as m:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                bases, offs = item
                m.ingest(bases, offs, n_reads)          # ctypes releases the GIL: the reader thread parses the next batch meanwhile
                n_reads += len(offs) - 1
                n_bases += len(bases)
            nodes = m.finalize()
            stats = m.stats()
            # edges on the GPU from the device-resident node table (the reference's single-threaded loop, src/main.rs:1017-1117);
            # the host copy of the list goes straight into the GFA writer
            raw = m.graph_edges(presimp, raw=True)
            edges = dict(n1=[0] * int(raw.n), presimp_removed=int(raw.presimp_removed))
            em = Emitter()
            em.write_gfa(prefix + ".gfa", nodes, raw)
    finally:
        stop.set()                          # error or not: release the reader (it closes the file) and wait for it
        th.join()
    if write_sequences:                              # second pass over the input for the node sequences
        def again():
            first = 0
            with Reader(path, strip_newlines) as r:
                for bases, offs in r.batches(batch_bases):
                    yield bases, offs, first
                    first += len(offs) - 1
        em.write_sequences(prefix + ".0.sequences", nodes, l, again())
    return dict(n_reads=n_reads, n_bases=n_bases, n_minimizers=stats["n_minimizers"], n_windows=stats["n_windows"],
                n_nodes_before=nodes["n_nodes_before"], n_nodes=nodes["n_nodes"], n_edges=len(edges["n1"]),
                presimp_removed=edges["presimp_removed"])


def run_multik(path, prefix, ks, l, density, min_abundance=2, reads_already_hpc=False, presimp=0.01, batch_bases=256 << 20,
               strip_newlines=False, device=-1):
    """One pass over the reads, one graph per k (the k sweep of the reference's utils/multik:69-78 without its contig
    feedback): the reads are sketched once, the sketches stay resident on the GPU, and every k only clears and refills the
    counting table (mdbg_reset) and rebuilds nodes and edges.  Writes <prefix>-k<k>.gfa; -> {k: counters}"""
    ks = list(ks)
    out = {}
    n_reads = n_bases = 0
    with Mdbg(ks[0], l, density, min_abundance, reads_already_hpc=reads_already_hpc, device=device) as m, Reader(path, strip_newlines) as r:
        for bases, offs in r.batches(batch_bases):
            m.ingest(bases, offs, n_reads)
            n_reads += len(offs) - 1
            n_bases += len(bases)
        em = Emitter()
        for i, k in enumerate(ks):
            if i:
                m.reset(k)                               # sketches stay; windows of the new k are inserted again
            nodes = m.finalize()
            raw = m.graph_edges(presimp, raw=True)
            em.write_gfa("%s-k%d.gfa" % (prefix, k), nodes, raw)
            st = m.stats()
            out[k] = dict(n_reads=n_reads, n_bases=n_bases, n_minimizers=st["n_minimizers"], n_windows=st["n_windows"],
                          n_nodes_before=nodes["n_nodes_before"], n_nodes=nodes["n_nodes"], n_edges=int(raw.n),
                          presimp_removed=int(raw.presimp_removed))
    return out
