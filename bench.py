#!/usr/bin/env python3
"""bench.py — Gbases/s ingested to the k-min-mer graph on synthetic HiFi-shaped reads (BASELINE.json metric).

A "step" = one pass of the hot path over reads already resident in HBM:
    reset -> [per batch: sketch (HPC + ntHash + density filter) -> k-min-mer windows -> counting table] -> finalized node table.
Workloads (--workload):
  fly    BASELINE.json configs[2], the single-GPU configuration and the default at N=1: synthetic D. melanogaster, 140 Mb genome @50x per GPU
         (~15 kb reads, 0.1 % errors), k=35 l=12 d=0.002 minabund=2, one batch per step.
  human  BASELINE.json configs[3], the default at N>1: synthetic human, 3 Gb genome @52x = 10.4 M reads = 156 Gbases, k=35 l=14 d=0.003, held as the
         EIGHT shards of the 8-GPU configuration (19.5 Gbases each).  The SAME data set at every N (strong scaling): rank r of N takes the shards
         [8 r / N, 8 (r + 1) / N) and pushes them through the multi-GPU layer as 8 / N batches per step; `--gpus 1 --workload human` streams all eight
         through one context — the N=1 point of the 1 -> 8 curve (the default N=1 line is the other workload).
At N>1 the table is partitioned by key range; the ranks exchange window lists + the sketch hashes they need over RCCL send/recv pairs inside
libmdbg_hip.so (include/mdbg_dist.h; default) or route the k-min-mer occurrences to their owner with one RCCL all-to-all per step (--dist-mode route;
rust_mdbg_amd/dist.py, fly workload only).  `python bench.py --gpus N` starts its N ranks itself (torch.distributed.run, one rank per GPU) and exits
non-zero when fewer than N GPUs are visible; under an external launcher WORLD_SIZE must equal --gpus; `n_gpus` in the line is the
number of ranks that took part in an all-reduce.  `--comm host` is a DRY RUN of the same multi-process path on fewer GPUs than ranks: the transfers are
staged through host memory and carried by gloo (rust_mdbg_amd/dist_c.py HostStagedComm), several ranks share a device, and the line says so — not RCCL, not a result.
At N>1 the line also carries `no_exchange_anchor`: the same ranks, each pushing its own shards through one local context right after the timed region
(no exchange, table not partitioned), and `n1_same_workload`: the committed N=1 line of the same workload.
The reads sit in HBM in the north star's layout, packed 2 bits per base (--input ascii: one byte per base, fly only); at N=1 the line also carries
`ascii_in`: the same steps fed ASCII, and the time of the device packer, so that the GPU and the CPU leg can be read from the same starting bytes,
and `scale_anchor_n1`: the human data set of the N>1 lines streamed through this one GPU after the timed region — the N=1 point of the 1 -> 8 curve in the same record as the headline.

Prints ONE JSON line (rank 0).  `roofline` refers to the dominant kernel (sketch_bs_kernel) as fed in the timed region and
is measured live with HIP events on the stream the kernel is launched on; `roofline_ascii` is the same kernel fed ASCII
(b_in = 0.25 vs 1.0 byte per base), measured right after the timed region; `cpu_baseline` is the CPU oracle
(a port of the reference's path) timed on a bounded sample of the same reads on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
# stages of the multi-GPU layer's host timers (mdbg_dist_stage_name), in the library's order
DIST_STAGES = ("sketch", "wait", "scatter", "commit", "owner_lists", "pack", "allgather", "reserve", "sync", "begin", "insert", "finalize_begin", "finalize_allreduce",
               "finalize_end", "position_fetch")
# what ONE rank of eight spends per step on configs[3], measured stage by stage on one GPU (profiles/r06_rank_w8.txt; m = measured, e = estimated: the two
# collectives have never run here) — printed with every N > 1 line of the human workload so that a result can be set against it the same day
BUDGET_8_RANKS_HUMAN = {
    "source": "DESIGN.md 3.4, profiles/r06_rank_w8.txt (one MI355X, a human shard of 19.5 Gbases, the seven peers' sketches made by the same GPU)",
    "sketch (tile kernel 4.9 + records, scan, gather)": [5.6, "m"], "owner_lists": [1.1, "m"], "pack": [0.9, "m"], "scatter": [0.9, "m"],
    "insert (own 1.0 + 40.6 M listed windows 4.8)": [5.8, "m"], "finalize_begin": [0.85, "m"], "finalize_allreduce (2 x 90 MB)": [1.3, "e"],
    "finalize_end incl. position_fetch 0.6": [1.3, "m + e"], "clear + host round trips": [0.9, "m"], "sum": [18.6, "= 4.5 x of 8 against 83.5 ms at N = 1"],
    "wire": "1.41 GB into a rank per step; chunk 0 travels under chunk 1's tile kernel, chunk 1 under the insertion of chunk 0's arrivals: exposed = `wait`"}
MULTIK = [10, 15, 20, 25, 30, 35, 40]      # utils/multik:69-78 (k from 10 to 40 in steps of 5)
HUMAN_SHARDS = 8            # --workload human: the data set is held as the eight shards of BASELINE.json configs[3]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "fly", "human"], default="auto", help="auto: fly (configs[2]) at N=1, human (configs[3]) at N>1")
    ap.add_argument("--genome-mb", type=float, default=None, help="fly: genome per GPU in Mb (default 140); human: the WHOLE genome in Mb (default 3000)")
    ap.add_argument("--coverage", type=float, default=None, help="default: 50 (fly), 52 (human)")
    ap.add_argument("-k", type=int, default=35)
    ap.add_argument("-l", type=int, default=None, help="default: 12 (fly), 14 (human)")
    ap.add_argument("--density", type=float, default=None, help="default: 0.002 (fly), 0.003 (human)")
    ap.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                    help="N>1 transport: RCCL (one GPU per rank), or host-staged over gloo: a DRY RUN of the multi-process path with several ranks per GPU, labelled as such")
    ap.add_argument("--minabund", type=int, default=2)
    ap.add_argument("--input", choices=["packed", "ascii"], default="packed", help="layout of the reads in HBM during the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--multik-exchange", choices=["whole", "segments"], default="whole", help="--multik at N > 1: what the ranks exchange (whole: once per sweep; segments: once per k)")
    ap.add_argument("--multik", action="store_true",
                    help="BASELINE.json configs[4]: a step = sketch ONCE (l=12 d=0.003), then the graph of every k in 10,15,..,40 from the resident sketches (mdbg_reset(new_k): the table is "
                         "cleared and refilled, nothing is sketched or — at N>1, where the ranks hold whole sketches — exchanged again); value counts every k's graph: bases x 7 / time")
    ap.add_argument("--watchdog-seconds", type=int, default=1500, help="a rank that is still running after this long dumps the tracebacks of all its threads to stderr and exits (a hang "
                    "in a collective would otherwise sit there until the caller's own limit, without a word); 0 = off")
    ap.add_argument("--no-scale-anchor", action="store_true", help="default N=1 run: skip the human data set's pass through this GPU after the timed region (scale_anchor_n1)")
    ap.add_argument("--no-anchor-oracle", action="store_true", help="default N=1 run: skip the CPU oracle's pass over one shard of the human data set (scale_anchor_n1.shard_vs_oracle, ~30 s)")
    ap.add_argument("--plain", action="store_true", help="only the warm-up and the timed steps (no ASCII legs, no edge stage, no CPU leg): for profiler runs, where every launch should be one of the timed kind")
    ap.add_argument("--force-dist", action="store_true", help="use the routed multi-GPU path even with one rank")
    ap.add_argument("--dist-mode", choices=["replicate", "route"], default="replicate",
                    help="multi-GPU mode: exchange of sketch hashes + key-partitioned table (default inside a node: 3.5x faster per rank and fewer bytes than records up to ~10 ranks, DESIGN.md 3.4), or all-to-all of k-min-mer records by key range (the north star's wording)")
    ap.add_argument("--dist-impl", choices=["py", "c"], default="c",
                    help="multi-GPU driver: the library's own C layer (include/mdbg_dist.h: direct RCCL send/recv groups, the product boundary; default), or "
                         "rust_mdbg_amd/dist.py over torch.distributed (the Python harness of the same protocol; the only driver of --dist-mode route)")
    ap.add_argument("--dist-exchange", choices=["segments", "whole"], default="segments",
                    help="C layer: what a round ships to a peer: its window list + only the hashes those windows need (default), or every sketch entire (mdbg_dist_set_exchange)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="multi-GPU: chunks per step; the exchange of chunk c overlaps the sketch of chunk c+1 (0 = 4 in replicate mode, 1 in route mode)")
    ap.add_argument("--profile-dist", action="store_true", help="print a per-stage wall-time breakdown of the routed path to stderr (adds syncs)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a.gpus, a.comm)            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run --nproc-per-node N with the same N" % (a.gpus, world))
    if a.dist_mode == "route":
        a.dist_impl = "py"                     # the record routing exists in the Python harness only
    if a.workload == "auto":
        a.workload = "human" if world > 1 and a.dist_impl == "c" else "fly"      # (the Python harness drives one batch per step: fly)
    human = a.workload == "human"
    if human and HUMAN_SHARDS % world:
        ap.error("--workload human is held as %d shards: --gpus must divide %d" % (HUMAN_SHARDS, HUMAN_SHARDS))
    if human and (a.input != "packed" or a.dist_impl != "c"):
        ap.error("--workload human: packed input and the C layer only")
    if a.multik:
        if not human or a.dist_impl != "c":
            ap.error("--multik runs on the human workload (configs[4]) and the C layer")
        if a.l is None: a.l = 12
        a.k = MULTIK[0]
        # the sweep's default at N > 1: every rank keeps every sketch entire, so a new k needs no new exchange (mdbg_dist_set_exchange); --multik-exchange segments: the
        # default exchange, and mdbg_dist_reset(k) exchanges the rounds again for every k (seven smaller exchanges instead of one large one)
        a.dist_exchange = a.multik_exchange
    if a.genome_mb is None: a.genome_mb = 3000.0 if human else 140.0
    if a.coverage is None: a.coverage = 52.0 if human else 50.0
    if a.l is None: a.l = 14 if human else 12
    if a.density is None: a.density = 0.003 if human else 0.002
    return a


def expected_graph(args, world, shard_reads, total_bases):
    """the graph this workload must produce (tests/golden/bench_counts.json), or None for a workload that has no recorded counts.  The human
    workload is the same data set at every N: its GLOBAL node count is pinned for every N, the other counts (per-rank stores) at N=1."""
    try:
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_counts.json")))
    except OSError:
        return None
    for w in ref["workloads"]:
        if (w.get("workload", "fly"), w["k"], w["l"], w["density"], w["minabund"], w["genome_mb"], w["coverage"]) != \
           (args.workload + ("-multik" if args.multik else ""), args.k, args.l, args.density, args.minabund, args.genome_mb, args.coverage):
            continue
        if args.workload == "human":
            if w["total_bases"] != total_bases:
                continue
            if args.multik:
                return {"nodes_per_k": w["graph"]["nodes_per_k"]}      # (global counts: the same at every N)
            # (N > 1: the global node count and the digest of the union of the partitions — the same SET as the one-GPU table — are pinned; the other counts are per-rank stores)
            return dict(w["graph"]) if world == 1 else {f: w["graph"][f] for f in ("nodes", "node_digest") if f in w["graph"]}
        if (w["n_gpus"], w["reads_per_gpu"], w["bases_per_gpu"]) == (world, shard_reads, total_bases):
            return w["graph"]
    return None


def n1_same_workload(args):
    """the committed N=1 line of the same workload (profiles/r*_human_n1.json: `bench.py --gpus 1 --workload human` on one MI355X), for the N>1 lines"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_human_n1.json")), reverse=True):
        try:
            j = json.load(open(p))
            c = j["config"]
            if (c["k"], c["l"], c["density"], c["minabund"], c["genome_mb"], c["coverage"]) == (args.k, args.l, args.density, args.minabund, args.genome_mb, args.coverage):
                return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "source": os.path.relpath(p, ROOT),
                        "what": "the same data set streamed through ONE GPU as %d batches (bench.py --gpus 1 --workload human), recorded earlier" % HUMAN_SHARDS}
        except Exception:
            continue
    return None


def self_launch(n, comm="rccl"):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run on this node and become
    that launcher.  Fewer than N visible GPUs is an error, not an N=1 run."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (n if comm == "rccl" else 1):
        raise SystemExit("bench.py: --gpus %d needs %d GPUs, %d visible on this node (one rank per GPU; RCCL does not run two ranks on one device)"
                         % (n, n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def pmc_traffic(bases_per_launch, args, fmt=None):
    """HBM bytes per launch of sketch_tile_kernel from the committed rocprofv3 PMC passes of this very command
    (profiles/summarize.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied).  PMC counters cannot
    be read from inside the process, so the figure is only reported when the workload of this run matches the profiled one."""
    import glob
    fmt = fmt or args.input
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            j = json.load(open(p))
            ref = json.load(open(p.replace("_pmc_traffic.json", "_bench.json")))
            k = next(v for n, v in j["kernels"].items() if n.startswith("sketch_bs_kernel<%d>" % args.l) or n.startswith("sketch_bs_kernel<%d," % args.l))
            c = ref["config"]
            same = (c["k"], c["l"], c["density"], c["minabund"], c.get("input_format")) == (args.k, args.l, args.density, args.minabund, args.input) and \
                abs(c["bases_per_gpu"] / ref["roofline"]["launches_per_step"] - bases_per_launch) < 1e-6 * bases_per_launch
            if same:
                by = j.get("tile_kernel_by_input") or {}
                return by.get(fmt, k["hbm_bytes_per_launch"] if not by else None), os.path.relpath(p, ROOT)
        except Exception:
            continue
    return None, None


def sq_counters(args):
    """VALU figures of the tile kernel from the committed SQ counter pass of this very command (profiles/r*_sq_counters.json;
    rocprofv3 --pmc cannot run inside the process): instructions per base and the share of the kernel's cycles in which a
    SIMD's VALU is busy."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")), reverse=True):
        try:
            j = json.load(open(p))
            c = j["config"]
            if (c["k"], c["l"], c["density"], c["minabund"], c["input_format"]) == (args.k, args.l, args.density, args.minabund, args.input):
                return dict(j["derived"], source=os.path.relpath(p, ROOT))
        except Exception:
            continue
    return None


def issue_roofline(args, roof, n_cus):
    """The tile kernel's instruction-issue ceiling next to its HBM one: VALU-pipe cycles the launch NEEDS (instruction counts from the SQ counters,
    split into full-rate and half-rate instructions by the ISA of the hot path, each at its measured issue cost: profiles/isa_issue.py ->
    profiles/r*_isa_issue.json) over the SIMD cycles it GETS (4 SIMDs x CUs x the live launch duration x 2.4 GHz)."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_issue.json")), reverse=True):
        try:
            j = json.load(open(p))
            c = j["config"]
            if (c["l"], c["density"], c["input_format"]) != (args.l, args.density, args.input):
                continue
            bases = roof["algorithmic_bytes_per_launch"] / roof["algorithmic_bytes_per_base"]
            need = j["valu_issue_cycles_per_base"] * bases
            have = 4.0 * n_cus * roof["avg_launch_ms"] * 1e-3 * 2.4e9
            allinst = j["all_instructions_per_launch"] / c["bases_per_launch"] * bases
            return {"valu_pipe": need / have, "valu_issue_cycles_needed": need, "simd_cycles_available": have, "clock_ghz": 2.4, "simds": 4 * n_cus,
                    "valu_instructions_per_launch": j["valu_instructions_per_base"] * bases, "half_rate_share": j["half_rate_share"],
                    "issue_cycles_full_rate": j["c_full"], "issue_cycles_half_rate": j["c_half"], "avg_issue_cycles_per_valu": j["avg_issue_cycles_per_valu"],
                    "all_instructions_per_simd_cycle": allinst / have,
                    "reads_as": "share of the SIMDs' cycles the launch's VALU instructions occupy at their measured issue cost; 1.0 = nothing but fewer or cheaper "
                                "instructions can make the kernel faster",
                    "source": os.path.relpath(p, ROOT)}
        except Exception:
            continue
    return None


def hex_digest(dg):
    """the order-free digest of a node table {(key, abundance)} (include/mdbg_hip.h, mdbg_nodes_digest; oracle/mdbg_oracle.cpp, orc_node_hash) as two hex words: sum, xor"""
    return None if dg is None else ["0x%016x" % (int(dg[0]) & 0xFFFFFFFFFFFFFFFF), "0x%016x" % (int(dg[1]) & 0xFFFFFFFFFFFFFFFF)]


def cpu_baseline(m_ctx, d_bases, d_off, n_reads, n_bases, args):
    """Times the CPU oracle on a bounded prefix of the same reads, one worker per host core."""
    import numpy as np
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    offs = m_ctx.to_host(d_off, (n_reads + 1) * 8, np.uint64)
    # calibrate on ~40 Mbases
    r0 = int(np.searchsorted(offs, 40_000_000, side="right"))
    r0 = max(1, min(r0, n_reads))
    b0 = m_ctx.to_host(d_bases, int(offs[r0]))
    t = time.perf_counter()
    O.count_threaded(b0, offs[:r0 + 1], args.k, args.l, args.density, args.minabund, threads=cores)
    dt = time.perf_counter() - t
    rate = float(offs[r0]) / dt
    target = min(float(n_bases), rate * args.cpu_seconds)
    r1 = int(np.searchsorted(offs, target, side="right"))
    r1 = max(r0, min(r1, n_reads))
    b1 = m_ctx.to_host(d_bases, int(offs[r1]))
    t = time.perf_counter()
    solid, wins, dg = O.count_digest_threaded(b1, offs[:r1 + 1], args.k, args.l, args.density, args.minabund, threads=cores)
    dt = time.perf_counter() - t
    # whole: the sample IS the workload of one step, so its node and window counts are the oracle's answer for the configuration the headline is
    # measured on (src/main.rs:926-932 prints the same counters) — main() compares them with the GPU's and prints no line when they differ
    return {"value": float(offs[r1]) / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "port",
            "nodes": int(solid), "windows": int(wins), "node_digest": hex_digest(dg), "whole_workload": bool(r1 == n_reads and int(offs[r1]) == int(n_bases)), "matches_gpu": None,
            "sample": "first %d reads (%.3f Gbases) of the same synthetic workload, %d threads, %.1f s; reads in RAM -> filtered node count; "
                      "the port sketches on all threads, deals the canonical k-min-mers into one bucket per thread by key hash and counts "
                      "every bucket on its own thread (no shared map, no serial merge)"
                      % (r1, float(offs[r1]) / 1e9, cores, dt)}


def api_stats_of(cdist):
    """mdbg_get_stats of the context inside an mdbg_dist"""
    import ctypes as C
    from rust_mdbg_amd import api
    s = api.Stats()
    L = api.load_library()
    L.mdbg_get_stats(C.c_void_p(L.mdbg_dist_ctx(cdist.h)), C.byref(s))
    return {f: getattr(s, f) for f, _ in api.Stats._fields_ if f != "reserved"}


def human_shards(m, torch, np, genome_mb, coverage, shard_ids):
    """the shards `shard_ids` of the human data set, generated on the device one after the other and kept packed: -> (batches, tensors that own them, reads per shard)"""
    genome_len = int(genome_mb * 1e6)
    shard_reads = int(genome_mb * 1e6 * coverage / 15000.0) // HUMAN_SHARDS
    keep, batches = [], []
    for g in shard_ids:
        d_bases, d_off0, nb = m.synth_reads_device(seed=1, genome_len=genome_len, n_reads=shard_reads, mean_len=15000, sd_len=1500,
                                                   min_len=8000, max_len=25000, err_ppm=1000, first_read=g * shard_reads)
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        offs = torch.from_numpy(m.to_host(d_off0, (shard_reads + 1) * 8, np.uint64).view(np.int64).copy()).cuda()
        torch.cuda.synchronize()
        assert m.pack_device(d_bases, nb, words.data_ptr()) == 0      # synthetic reads are pure ACGT
        m.sync()
        keep += [words, offs]
        batches.append((words.data_ptr(), offs.data_ptr(), shard_reads, nb, g * shard_reads))
    return batches, keep, shard_reads, (d_bases, d_off0)


def scale_anchor_n1(R, torch, np, device_index, minabund, oracle_shard=False):
    """The N=1 point of the 1 -> 8 curve measured in the DEFAULT N=1 run as well: the multi-GPU lines run BASELINE configs[3] (the same 156 Gbases at every N), the
    default N=1 line configs[2] — so the human data set goes through this one GPU once more here, after the timed region (about 3 s), and the curve's anchor sits in
    the same driver record as the headline."""
    k, l, d, gm, cov = 35, 14, 0.003, 3000.0, 52.0
    # The blocks the library and torch keep for reuse go back to the runtime first, so that the anchor's 13-GB table and 6-GB store are fresh allocations as in a run of
    # their own.  (The pass varies by +-1.3 % from process to process on one box — five fresh processes: 84.1 / 85.0 / 86.0 / 86.1 / 86.3 ms, scratch/anchor_alone.py —,
    # which is more than what the legs in front of it cost, if anything: where the table lands is drawn anew with every allocation.)
    try:
        from rust_mdbg_amd import api as _api
        _api.release_cached_memory()
        torch.cuda.empty_cache()
    except Exception:
        pass
    with R.Mdbg(k, l, d, minabund, device=device_index) as mh:
        batches, keep, shard_reads, last_ascii = human_shards(mh, torch, np, gm, cov, range(HUMAN_SHARDS))

        last = {}

        def one():
            mh.reset(0)
            for (b_in, b_off, b_reads, b_bases, b_first) in batches:
                mh.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first)
            last["nd"] = mh.finalize_device()
            return int(last["nd"].n)
        mh.set_timing(1)                          # as in the timed region of the human workload: the tile kernel's events only
        one()                                     # sizes the store and the table
        one()                                     # (the table's last growth may fall into the second pass)
        mh.sync()
        t1 = time.perf_counter()
        steps = 4
        for _ in range(steps):
            nodes = one()
        mh.sync()
        ms = (time.perf_counter() - t1) / steps * 1e3
        st = mh.stats()
        total = sum(b[3] for b in batches)
        digest = hex_digest(mh.nodes_digest(last["nd"]))
        shard_check = None
        if oracle_shard and "scale_anchor_oracle" in os.environ.get("MDBG_BENCH_FAIL_SIDE", "").split(","):
            shard_check = {"error": "forced by MDBG_BENCH_FAIL_SIDE"}
        elif oracle_shard:
            try:
                # One shard of configs[3] (19.5 Gbases: the batch a rank of eight ingests per step) through the GPU alone and through the CPU oracle: node count, window count and
                # the node-set digest must agree — the human PARAMETERS at full batch size against the oracle in the driver's own record (the whole data set would be 8 x this).
                from oracle import oracle as O
                g = HUMAN_SHARDS - 1                   # the shard generated last: its ASCII is still in the context's buffer
                b_in, b_off, b_reads, b_bases, b_first = batches[g]
                d_bases, d_off0 = last_ascii
                mh.reset(0)
                mh.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first)
                nd1 = mh.finalize_device()
                st1 = mh.stats()
                gpu1 = (int(nd1.n), int(st1["n_windows"]), hex_digest(mh.nodes_digest(nd1)))
                offs = mh.to_host(d_off0, (b_reads + 1) * 8, np.uint64)
                bases = mh.to_host(d_bases, int(offs[-1]))
                cores = os.cpu_count() or 1
                t2 = time.perf_counter()
                solid, wins, dg = O.count_digest_threaded(bases, offs, k, l, d, minabund, threads=cores)
                dt2 = time.perf_counter() - t2
                del bases
                shard_check = {"what": "shard %d of the data set alone (%.1f Gbases): GPU node table vs the CPU oracle on the same reads" % (g, b_bases / 1e9), "nodes": int(solid), "windows": int(wins),
                               "node_digest": hex_digest(dg), "gpu_nodes": gpu1[0], "gpu_windows": gpu1[1], "gpu_node_digest": gpu1[2], "cpu_seconds": dt2, "cores": cores,
                               "cpu_gbases_per_s": b_bases / dt2 / 1e9, "matches_gpu": bool((int(solid), int(wins), hex_digest(dg)) == gpu1)}
            except Exception as ex:          # (a side measurement of a side measurement: the anchor's timing above must not be lost to it)
                shard_check = {"error": repr(ex)[:300]}
        del keep
    return {"what": "BASELINE.json configs[3] (synthetic human 3 Gb @52x, k=35 l=14 d=0.003): the workload of the N>1 lines of this script, streamed through THIS one GPU as "
                    "%d batches per step (= bench.py --gpus 1 --workload human), after the timed region" % HUMAN_SHARDS,
            "value": total / ms / 1e6, "unit": "Gbases/s", "ms_per_step": ms, "steps": steps, "total_bases": total,
            "graph": {"minimizers": st["n_minimizers"], "windows": st["n_windows"], "distinct": st["n_distinct"], "nodes": nodes, "node_digest": digest},
            "shard_vs_oracle": shard_check}


def main():
    args = parse()
    if args.watchdog_seconds > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)      # (cancelled by the process's end)
    # stdout carries exactly one JSON line: native libraries (RCCL prints a version banner) write to file descriptor 1
    # directly, so fd 1 is pointed at stderr for the whole run and the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    host_comm = args.comm == "host" and world > 1
    n_dev = torch.cuda.device_count()
    if n_dev < world and not host_comm:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (one rank per GPU)" % (world, n_dev))
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    dist = None
    routed = world > 1 or args.force_dist
    human = args.workload == "human"
    if routed:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        if host_comm:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
    red_dev = "cpu" if host_comm else "cuda"          # where the scalars of the bench's own all-reduces live

    def allreduce(vals, dtype, op=None):
        t = torch.tensor(vals, device=red_dev, dtype=dtype)
        dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM)
        return t.tolist()
    import rust_mdbg_amd as R

    packed = args.input == "packed"
    m = R.Mdbg(args.k, args.l, args.density, args.minabund, device=device_index)
    # ---- the reads of this rank: a list of batches (device pointers) --------------------------------------------------------------
    keep = []                                         # tensors that own the batches' memory
    batches = []                                      # (d_in, d_off, n_reads, n_bases, first_ordinal)
    d_bases = d_off0 = None                           # ASCII of the (last generated) batch: owned by the context
    pack_ms = None
    if human:
        per_rank = HUMAN_SHARDS // world
        batches, keep, shard_reads, (d_bases, d_off0) = human_shards(m, torch, np, args.genome_mb, args.coverage, range(rank * per_rank, (rank + 1) * per_rank))
        reads_per_gpu = shard_reads * per_rank
    else:
        genome_len = int(args.genome_mb * 1e6) * world            # weak scaling: coverage constant, genome grows with N
        reads_per_gpu = shard_reads = int(args.genome_mb * 1e6 * args.coverage / 15000.0)
        d_bases, d_off0, nb = m.synth_reads_device(seed=1, genome_len=genome_len, n_reads=reads_per_gpu, mean_len=15000, sd_len=1500,
                                                   min_len=8000, max_len=25000, err_ppm=1000, first_read=rank * reads_per_gpu)
        d_in = d_bases
        if packed:          # outside the timed region: the batch as the host packer (mdbg_pack_reads) would have delivered it
            words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
            exc = (torch.zeros(64, dtype=torch.int64, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda"))
            torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
            for _ in range(2):            # (second pass timed: what the device packer costs on top of a step that starts from ASCII)
                m.sync()
                t1 = time.perf_counter()
                assert m.pack_device(d_bases, nb, words.data_ptr(), exc[0].data_ptr(), exc[1].data_ptr(), 64) == 0      # synthetic reads are pure ACGT
                m.sync()
                pack_ms = (time.perf_counter() - t1) * 1e3
            keep += [words, exc]
            d_in = words.data_ptr()
        batches.append((d_in, d_off0, reads_per_gpu, nb, rank * reads_per_gpu))
    n_bases = sum(b[3] for b in batches)

    cdist = None
    replicate = args.dist_mode == "replicate"
    n_chunks = 1
    if routed and args.dist_impl == "c":
        from rust_mdbg_amd import dist_c
        cdist = dist_c.DistMdbg(args.k, args.l, args.density, args.minabund, rank, world, dist, device=device_index, transport="host" if host_comm else "rccl")
        n_chunks = args.chunks if args.chunks > 0 else (4 if args.dist_exchange == "whole" else 2)      # segments: a rank's share of a round is a few tens of MB per link
        cdist.set_pipeline(n_chunks)          # the exchange of chunk i overlaps the tile kernel of chunk i+1 (mdbg_dist_set_pipeline)
        cdist.set_exchange(args.dist_exchange == "whole")
    if routed and cdist is None:
        from rust_mdbg_amd import dist as D
        dev = torch.device("cuda", device_index)
        n_chunks = args.chunks if args.chunks > 0 else (4 if replicate else 1)
        chunked = n_chunks > 1 and not args.profile_dist
        mt = R.Mdbg(args.k, args.l, args.density, args.minabund, device=device_index) if chunked and not replicate else None     # owner-side context
        engine = D.GpuEngine(m, torch, dev, table=mt)
        engine.packed = packed
        comm = D.TorchDistComm(dist, torch, dev)
        runner = D.ReplicatedMdbg(engine, comm, torch) if replicate else D.DistributedMdbg(engine, comm, torch, profile=args.profile_dist)
        d_in, d_off, _, _, first_ordinal = batches[0]
        if chunked:
            plan = D.plan_chunks(m.to_host(d_off, (reads_per_gpu + 1) * 8, np.uint64), n_chunks, keep_empty=replicate)
            offs_t = engine._view(d_off, (reads_per_gpu + 1,))

    last_nd = {}                                      # the device node table of the last step (local context / multi-GPU layer): what the digest below is taken of

    def local_step(ascii_in=False):
        """the whole hot path on this rank's own context (no exchange): reset -> batches -> finalize"""
        m.reset(0)
        for (b_in, b_off, b_reads, b_bases, b_first) in batches:
            if ascii_in:
                m.ingest_device(d_bases, b_off, b_reads, b_bases, b_first)
            elif packed:
                m.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first)
            else:
                m.ingest_device(b_in, b_off, b_reads, b_bases, b_first)
        last_nd["local"] = m.finalize_device()
        return last_nd["local"].n

    per_k = []                                        # --multik: (k, global nodes) of the last sweep
    sweep_stats = {}                                  # --multik: the context's stats right after the sweep's ingest (every mdbg_reset restarts the timers)

    def step():
        if args.multik:
            ctx = cdist if cdist is not None else m
            ctx.reset(0)
            ctx.reset(MULTIK[0])                      # (the sweep before left the context at the last k; nothing is resident: only k changes)
            for (b_in, b_off, b_reads, b_bases, b_first) in batches:
                (ctx.ingest_packed_device if packed else ctx.ingest_device)(b_in, b_off, b_reads, b_bases, b_first)
            sweep_stats.clear()
            sweep_stats.update(api_stats_of(cdist) if cdist is not None else m.stats())
            per_k.clear()
            for k in MULTIK:
                if k != MULTIK[0]:
                    ctx.reset(k)                      # clears the table and inserts the windows of the new k from the resident sketches
                if cdist is not None:
                    nd, _, ng = cdist.finalize()
                    cdist.last_local = int(nd.n)
                else:
                    ng = int(m.finalize_device().n)
                per_k.append((k, int(ng)))
            return per_k[-1][1]
        if cdist is not None:
            cdist.reset(0)
            for (b_in, b_off, b_reads, b_bases, b_first) in batches:
                if packed:
                    cdist.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first)
                else:
                    cdist.ingest_device(b_in, b_off, b_reads, b_bases, b_first)
            nd, _, ng = cdist.finalize()
            cdist.last_local = int(nd.n)
            last_nd["dist"] = nd
            return ng
        if not routed:
            return local_step()
        runner.reset() if replicate else engine.reset()
        if chunked:
            runner.ingest_device_chunked(d_in, offs_t, plan, first_ordinal)
        else:
            runner.ingest_device(d_in, d_off, reads_per_gpu, n_bases, first_ordinal)
        return runner.finalize_device_count()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        m.sync()

    # The timed steps record the events around the tile kernel only (mdbg_set_timing(1): what `roofline` is made of); the events around the stages (~4 microseconds of
    # stream time each, six per batch + two per finalize) are switched on for ONE further step after the timed region, which is what stage_ms_last_step reports.
    timed_level = 2 if (os.environ.get("MDBG_BENCH_STAGE_EVENTS") or args.multik or (routed and cdist is None)) else 1      # (the variable: A/B switch — the stage events inside the
                                                                                                                            # timed region, as until round 5; the sweep and the Python harness keep them)
    for ctx_ in ([m] + ([cdist] if cdist is not None else [])):
        ctx_.set_timing(timed_level)
    for _ in range(args.warmup):
        step()
    fence()
    if routed and cdist is None and not replicate:
        runner.times = {}
    # the tile kernel's HIP-event time of EVERY timed step (the context's timers restart with each reset): host-side reads of a struct, no device sync
    tile_acc = {"ms_sketch_tile": 0.0, "n_sketch_tile_launches": 0, "n_sketch_tile_bases": 0}
    if cdist is not None:
        cdist.stage_ms(reset=True)          # the layer's host timers per stage restart with the timed region (mdbg_dist_stage_ms)
    t0 = time.perf_counter()
    n_nodes = 0
    for _ in range(args.steps):
        n_nodes = step()
        sti = sweep_stats if args.multik else (api_stats_of(cdist) if cdist is not None else m.stats())
        for f in tile_acc:
            tile_acc[f] += sti[f]
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = float(allreduce([dt], torch.float64, dist.ReduceOp.MAX)[0])
        total_bases, n_ranks = (int(v) for v in allreduce([n_bases, 1], torch.int64))      # [bases, 1]: the second sum = ranks the collective actually reached
    else:
        total_bases, n_ranks = n_bases, 1
    if n_ranks != args.gpus:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) took part in the all-reduce" % (args.gpus, n_ranks))
    # outside the timed region: the order-free digest of the node table the last timed step left on the device — of every rank's partition at N > 1, folded over the
    # ranks (sum mod 2^64, XOR): the digest of the UNION of the partitions, comparable with the one-GPU table's and with the CPU oracle's (cpu_baseline)
    node_digest = None
    if not args.multik and (cdist is not None or not routed):
        if os.environ.get("MDBG_BENCH_CORRUPT") == "abundance" and not routed and int(last_nd["local"].n):      # (test hook: one abundance of the device table + 1 — the line must be refused)
            import ctypes
            addr = ctypes.cast(last_nd["local"].abundance, ctypes.c_void_p).value
            a0 = m.to_host(addr, 2, np.uint16)
            a0[0] = np.uint16((int(a0[0]) + 1) & 0xFFFF)
            assert m.L.mdbg_copy_to_device(m.h, ctypes.c_void_p(addr), a0.ctypes.data, 2) == 0
        dg = cdist.nodes_digest(last_nd["dist"]) if cdist is not None else m.nodes_digest(last_nd["local"])
        if dist is not None:
            mine = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in dg], device=red_dev, dtype=torch.int64)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            sm, xr = 0, 0
            for t in parts:
                a_, b_ = (int(v) & 0xFFFFFFFFFFFFFFFF for v in t.tolist())
                sm = (sm + a_) & 0xFFFFFFFFFFFFFFFF; xr ^= b_
            dg = (sm, xr)
        node_digest = hex_digest(dg)
    consistent = None
    if routed:               # outside the timed region: the ranks' partitions must add up to the global node count
        loc = int(allreduce([(cdist if cdist is not None else runner).last_local], torch.int64)[0])
        consistent = bool(loc == int(n_nodes))
    exchange = None
    if cdist is not None:    # what went over the links in the last step, and how even the partition is (mdbg_dist_traffic; outside the timed region)
        try:
            b_in, b_out, n_q = cdist.traffic()
        except Exception as ex:          # (a side measurement; every rank still joins the all-reduces below)
            print("bench.py: mdbg_dist_traffic failed on rank %d: %r" % (rank, ex), file=sys.stderr)
            b_in = b_out = n_q = 0
        tmax = allreduce([b_in, int(cdist.last_local)], torch.int64, dist.ReduceOp.MAX)
        tsum = allreduce([b_in, int(cdist.last_local)], torch.int64)
        # the layer's host time per stage of the timed steps (mdbg_dist_stage_ms): slowest rank per stage and mean over the ranks, per step — what the budget of DESIGN.md
        # 3.4 is compared with, stage by stage
        layer = None
        try:
            stg, n_rounds = cdist.stage_ms()
        except Exception as ex:
            print("bench.py: mdbg_dist_stage_ms failed on rank %d: %r" % (rank, ex), file=sys.stderr)
            stg, n_rounds = {}, 0
        names = list(DIST_STAGES)
        vals = [float(stg.get(nm, 0.0)) / max(1, args.steps) for nm in names]
        smax = allreduce(vals, torch.float64, dist.ReduceOp.MAX)
        ssum = allreduce(vals, torch.float64)
        layer = {"what": "host milliseconds per step and stage of the multi-GPU layer inside the timed region (the stages end in stream syncs or are host work); position_fetch is "
                         "part of finalize_end; sketch includes the tile kernel of the rank's own chunks",
                 "rounds_per_step": n_rounds / max(1, args.steps),
                 "slowest_rank": {nm: round(float(v), 3) for nm, v in zip(names, smax)}, "mean": {nm: round(float(v) / world, 3) for nm, v in zip(names, ssum)}}
        exchange = {"mode": args.dist_exchange, "layer_ms_per_step": layer, "budget_ms_per_step_8_ranks_human": BUDGET_8_RANKS_HUMAN if args.workload == "human" else None, "chunks_per_step": n_chunks * len(batches), "bytes_in_busiest_rank_per_step": int(tmax[0]), "bytes_in_mean_per_step": float(tsum[0]) / world,
                    "nodes_busiest_rank_over_mean": (float(tmax[1]) * world / float(tsum[1])) if int(tsum[1]) else None,
                    "transport": "host-staged over gloo (--comm host: DRY RUN, not RCCL)" if host_comm else "RCCL (grouped ncclSend / ncclRecv inside libmdbg_hip.so)"}
    anchor = None
    if cdist is not None and not args.multik:    # outside the timed region: what the same ranks do WITHOUT the exchange — every rank pushes its own shards through one local context
        # (table not partitioned).  N x this rate is the ceiling on this node for this split of the data.
        t_loc = -1.0
        try:
            if "no_exchange_anchor" in os.environ.get("MDBG_BENCH_FAIL_SIDE", "").split(","):
                raise RuntimeError("forced by MDBG_BENCH_FAIL_SIDE")
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); m.sync()
                t1 = time.perf_counter()
                local_step()
                m.sync()
                ts.append(time.perf_counter() - t1)
            t_loc = min(ts[1:])          # the first pass sizes the store and the table
            m.reset(0)
        except Exception as ex:          # (the line above it is the result; a failure here must not lose it — but every rank still joins the all-reduce)
            print("bench.py: local anchor pass failed on rank %d: %r" % (rank, ex), file=sys.stderr)
        ta = allreduce([t_loc, -t_loc], torch.float64, dist.ReduceOp.MAX)
        t_max, t_min = float(ta[0]), -float(ta[1])
        if t_min > 0:
            anchor = {"what": "the same %d rank(s), each pushing its own %d batch(es) through one local context: no exchange, table not partitioned (after the timed region; "
                              "slowest rank)" % (world, len(batches)), "value": total_bases / t_max / 1e9, "unit": "Gbases/s", "ms_per_step": t_max * 1e3}
    # one more step with the stage timers on (outside the timed region): its stage times are the line's stage_ms_last_step
    for ctx_ in ([m] + ([cdist] if cdist is not None else [])):
        ctx_.set_timing(2)
    if not args.multik and (cdist is not None or not routed):
        step()
        fence()
    st = m.stats()          # stats of the last step only (reset clears the timers)
    if cdist is not None:
        m_stats = api_stats_of(cdist)
        for f in ("n_minimizers", "n_windows", "n_distinct", "table_capacity", "n_slow_tiles", "n_tiles", "ms_sketch", "ms_sketch_tile", "ms_insert", "ms_finalize",
                  "n_sketch_tile_launches", "n_sketch_tile_bases", "n_bases"):
            st[f] = m_stats[f]
    if args.multik:          # the sketch side of the line comes from the sweep's ingest, the table side from its last k
        for f in ("ms_sketch", "ms_sketch_tile", "n_sketch_tile_launches", "n_sketch_tile_bases", "n_tiles", "n_slow_tiles"):
            st[f] = sweep_stats[f]
    if routed and cdist is None and engine.tm is not m:
        st2 = engine.tm.stats()
        st["n_distinct"], st["table_capacity"] = st2["n_distinct"], st2["table_capacity"]
        st["ms_insert"] += st2["ms_insert"]
        st["ms_finalize"] += st2["ms_finalize"]
    if routed and cdist is None and not replicate and args.profile_dist and rank == 0:
        n = args.steps
        print("[dist profile, ms per step] " + ", ".join("%s=%.2f" % (k, v / n) for k, v in runner.times.items()), file=sys.stderr)
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        # --multik: the reads are sketched ONCE per step, so `value` stays the BASELINE metric (bases ingested / time, here: to SEVEN graphs); the figure that
        # counts every graph's bases sits beside it as multik_graph_gbases_per_s and cannot be read as the metric
        value = total_bases * args.steps / dt / 1e9
        multik_graph_rate = total_bases * args.steps * len(MULTIK) / dt / 1e9 if args.multik else None
        mins_per_base = st["n_minimizers"] / max(1, st["n_bases"])

        def roofline(stt, b_in, fmt, m_per_base=None):
            """algorithmic bytes per raw base of the sketch kernel = b_in + 12 m (input + u64 hash + u32 position)"""
            if not stt["n_sketch_tile_launches"]:
                return None
            per_base = b_in + 12.0 * (mins_per_base if m_per_base is None else m_per_base)
            alg = stt["n_sketch_tile_bases"] * per_base
            avg_ms = stt["ms_sketch_tile"] / stt["n_sketch_tile_launches"]
            ach = alg / stt["n_sketch_tile_launches"] / (avg_ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "sketch_bs_kernel<%d>" % args.l, "input_format": fmt, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": alg / stt["n_sketch_tile_launches"], "launches_per_step": stt["n_sketch_tile_launches"],
                    "avg_launch_ms": avg_ms, "algorithmic_bytes_per_base": per_base,
                    "kernel_gbases_per_s": stt["n_sketch_tile_bases"] / (stt["ms_sketch_tile"] * 1e-3) / 1e9}
        roof = roofline(dict(st, **tile_acc), 0.25 if packed else 1.0, args.input)          # the kernel's average launch duration over all timed steps
        if roof:
            roof["launches_per_step"] = tile_acc["n_sketch_tile_launches"] / max(1, args.steps); roof["launches_timed"] = tile_acc["n_sketch_tile_launches"]
        side_errors = {}

        def side(name, fn, default=None):
            """a side measurement (recorded profiles, passes after the timed region): a failure in one is reported in the line under side_errors, the headline is not
            lost to it.  (A parity mismatch raises SystemExit, which passes through: no line is printed then.)"""
            try:
                if name in os.environ.get("MDBG_BENCH_FAIL_SIDE", "").split(","):      # (tests: every side measurement can be made to fail)
                    raise RuntimeError("forced by MDBG_BENCH_FAIL_SIDE")
                return fn()
            except Exception as ex:
                side_errors[name] = repr(ex)[:300]
                print("bench.py: side measurement %s failed: %r" % (name, ex), file=sys.stderr)
                return default
        if roof:
            roof["traffic"], roof["traffic_source"] = side("pmc_traffic", lambda: pmc_traffic(st["n_sketch_tile_bases"] / st["n_sketch_tile_launches"], args), (None, None))
            roof["issue"] = side("issue_roofline", lambda: issue_roofline(args, roof, torch.cuda.get_device_properties(device_index).multi_processor_count)) if packed else None
            sq = side("sq_counters", lambda: sq_counters(args))
            if sq:
                roof.update(valu_lane_ops_per_base=sq.get("valu_lane_ops_per_base"), wave_time_split=sq.get("wave_time_split"), sq_source=sq.get("source"))
            if packed:
                roof["note"] = ("b_in = 0.25 B/base (2-bit packed input, the north-star layout): the kernel is bound by instruction issue (integer VALU, LDS, scalar), "
                                "not by HBM (valu_lane_ops_per_base%s); the same kernel on the b_in = 1.0 accounting is in roofline_ascii"
                                % ("; traffic = %.2fx the algorithmic bytes" % (roof["traffic"] / roof["algorithmic_bytes_per_launch"]) if roof["traffic"] else ""))
        roof_ascii = ascii_in = None

        def ascii_leg():
            # the same kernel fed one byte per base: the other accounting of SURVEY.md 8d, measured live
            for _ in range(2):
                m.reset(0)
                m.sketch_device(d_bases, d_off0, reads_per_gpu, n_bases, rank * reads_per_gpu)
            sta = m.stats()
            ra = roofline(sta, 1.0, "ascii")
            if ra:
                ra["traffic"], ra["traffic_source"] = pmc_traffic(sta["n_sketch_tile_bases"] / sta["n_sketch_tile_launches"], args, "ascii")
            # the same STEPS fed ASCII (both legs start from ASCII, concatenated + offsets): the timed region above starts from the packed layout,
            # which the device packer produces in pack_ms
            local_step(ascii_in=True)
            m.sync()
            t1 = time.perf_counter()
            na = max(3, min(10, args.steps))
            for _ in range(na):
                local_step(ascii_in=True)
            m.sync()
            ms_a = (time.perf_counter() - t1) / na * 1e3
            return ra, {"ms_per_step": ms_a, "value": n_bases / ms_a / 1e6, "unit": "Gbases/s", "steps": na, "pack_ms": pack_ms,
                        "value_pack_then_packed": n_bases / (pack_ms + ms_step) / 1e6 if pack_ms else None,
                        "what": "the same steps with the reads resident as ASCII (one byte per base): the tile kernel converts on the fly; pack_ms = mdbg_pack_device "
                                "(ASCII -> 2-bit planes) alone; value_pack_then_packed = bases / (pack_ms + ms_per_step of the timed region)"}
        def hpc_leg():
            # The same launch with reads_already_hpc = 1 (src/read.rs:186-192, --skiphpc): the condition the reference's published timings were taken under ("reads and
            # assemblies were homopolymer-compressed in those experiments", README.md:134).  The tile kernel then skips its phase 2 (keep masks, compaction, dense
            # stream: a third of its instructions) and hashes the text as it is — every raw position is a dense one, so phase 3 walks a stream 1 / 0.75 as long.
            with R.Mdbg(args.k, args.l, args.density, args.minabund, reads_already_hpc=True, device=device_index) as mh:
                b_in, b_off, b_reads, b_bases, b_first = batches[0]
                for _ in range(3):
                    mh.reset(0)
                    mh.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first, sketch_only=True)
                sth = mh.stats()
            rh = roofline(sth, 0.25, "packed, reads_already_hpc", sth["n_minimizers"] / max(1, sth["n_bases"]))
            if rh:
                rh["minimizers_per_base"] = sth["n_minimizers"] / max(1, sth["n_bases"])
                rh["vs_timed_kernel"] = rh["avg_launch_ms"] / roof["avg_launch_ms"] if roof else None
                rh["what"] = ("the tile kernel of the timed region on the same packed reads with reads_already_hpc = 1 (no homopolymer compression: phase 2 is skipped, the "
                              "dense stream is the raw text); what the compaction costs, as a measurement")
            return rh
        def syncmer_leg():
            # SURVEY.md 8 row f4 (--syncmers -s, src/read.rs:215-352; the reference's default s = 4, src/main.rs:438): the same packed reads sketched under the syncmer scheme,
            # l = 12 s = 4 d = 0.05 — the tile kernel's syncmer instantiation (phase 3: the window-minimum machine as a scan), HIP events around its launches
            with R.Mdbg(args.k, 12, 0.05, args.minabund, syncmer_s=4, device=device_index) as ms_:
                b_in, b_off, b_reads, b_bases, b_first = batches[0]
                for _ in range(3):
                    ms_.reset(0)
                    ms_.ingest_packed_device(b_in, b_off, b_reads, b_bases, b_first, sketch_only=True)
                sts = ms_.stats()
            if not sts["n_sketch_tile_launches"]:
                return None
            ms_k = sts["ms_sketch_tile"] / sts["n_sketch_tile_launches"]
            return {"l": 12, "s": 4, "density": 0.05, "kernel_ms": ms_k, "kernel_gbases_per_s": sts["n_sketch_tile_bases"] / sts["n_sketch_tile_launches"] / (ms_k * 1e-3) / 1e9,
                    "sketch_ms": sts["ms_sketch"], "minimizers_per_base": sts["n_minimizers"] / max(1, sts["n_bases"]),
                    "what": "the batch of the timed region sketched under --syncmers (l = 12, s = 4, d = 0.05): tile kernel alone, and the whole sketch stage (kernel + scan + gather)"}
        roof_hpc = sync_leg = None
        if packed and not routed and not human and not args.plain:
            roof_ascii, ascii_in = side("ascii_in", ascii_leg, (None, None))
            roof_hpc = side("roofline_hpc_input", hpc_leg)
            sync_leg = side("syncmers", syncmer_leg)
            local_step()      # the table the edge stage and the baseline below refer to (not a side measurement: the counts below come from it)

        def edge_leg():
            m.graph_edges_device(0.01)
            t1 = time.perf_counter()
            e = m.graph_edges_device(0.01)
            return {"ms": (time.perf_counter() - t1) * 1e3, "n_edges": int(e.n), "presimp_removed": int(e.presimp_removed)}
        edges = None
        if not routed and not human and not args.plain:               # outside the timed region: the edge stage that follows the hot path (device-resident in, device-resident out)
            edges = side("edges_after_timed_region", edge_leg)
        cpu = None
        if args.cpu_seconds > 0 and world == 1 and not args.plain:          # rank 0 at N=1 only: at N>1 the other ranks would wait for it
            # (human: d_bases / d_off0 hold the ASCII of the last shard generated — a sample of the same data set)
            cpu = side("cpu_baseline", lambda: cpu_baseline(m, d_bases, d_off0, shard_reads, batches[-1][3], args))
        anchor1 = None
        if world == 1 and not routed and not human and not args.plain and not args.no_scale_anchor and (args.genome_mb, args.coverage, args.l, args.density) == (140.0, 50.0, 12, 0.002):
            try:
                anchor1 = scale_anchor_n1(R, torch, np, device_index, args.minabund, oracle_shard=args.cpu_seconds > 0 and not args.no_anchor_oracle)
            except Exception as ex:      # (a side measurement: the headline above it must not be lost to it — e.g. a device shared with somebody else's 200 GB)
                anchor1 = {"error": repr(ex)[:300]}
            if (anchor1.get("shard_vs_oracle") or {}).get("matches_gpu") is False:
                raise SystemExit("bench.py: one shard of configs[3]: the oracle finds %r, the GPU %r: no line printed" % (
                    [anchor1["shard_vs_oracle"][f] for f in ("nodes", "windows", "node_digest")], [anchor1["shard_vs_oracle"][f] for f in ("gpu_nodes", "gpu_windows", "gpu_node_digest")]))
            if "graph" in anchor1:
                w3 = [w for w in json.load(open(os.path.join(ROOT, "tests", "golden", "bench_counts.json")))["workloads"] if w.get("workload") == "human" and w["total_bases"] == anchor1["total_bases"]]
                anchor1["checked_against_recorded_counts"] = bool(w3)
                if w3 and args.minabund == 2 and any(anchor1["graph"][f] != w3[0]["graph"][f] for f in w3[0]["graph"]):
                    raise SystemExit("bench.py: the graph of the scale anchor %r differs from the recorded one %r: no line printed" % (anchor1["graph"], w3[0]["graph"]))
        n1_same = None
        if human and world > 1:
            # the N=1 point of the same curve: measured in THIS run on rank 0's GPU when the ranks have a device each (the whole data set streamed through one local context
            # while the other ranks wait: the same box, the same day, the same code as the N>1 value above it); a dry run on a shared device, or a failure, falls back to
            # the committed N=1 line of the same workload (another box: the pool's boxes differ by a few per cent)
            def same_run():
                if host_comm or args.multik or (args.k, args.l, args.density, args.genome_mb, args.coverage) != (35, 14, 0.003, 3000.0, 52.0):
                    return None          # (a dry run on a shared device, or not the configs[3] parameters: not measured here)
                a1 = scale_anchor_n1(R, torch, np, device_index, args.minabund)
                return {"value": a1["value"], "unit": a1["unit"], "ms_per_step": a1["ms_per_step"], "source": "this run, rank 0's GPU, after the timed region", "graph": a1["graph"],
                        "what": "the same data set streamed through ONE GPU as %d batches (= bench.py --gpus 1 --workload human)" % HUMAN_SHARDS}
            n1_same = side("n1_same_workload_same_run", same_run)
            if n1_same is None:
                n1_same = side("n1_same_workload", lambda: n1_same_workload(args))
        graph = {"minimizers": st["n_minimizers"], "windows": st["n_windows"], "distinct": st["n_distinct"], "nodes": int(n_nodes), "node_digest": node_digest}
        if cpu is not None and cpu.get("whole_workload") and not human and not routed and not args.multik and not os.environ.get("MDBG_STOP_PHASE"):
            # the oracle has just counted the very reads the timed steps ingested: a full-size parity check of the headline configuration in every run
            # ... and the node SETS: the oracle's digest over its (key, abundance) pairs against the digest of the device table of the last timed step
            cpu["matches_gpu"] = bool(cpu["nodes"] == graph["nodes"] and cpu["windows"] == graph["windows"] and cpu["node_digest"] == node_digest)
            if not cpu["matches_gpu"]:
                raise SystemExit("bench.py: the oracle finds %d nodes / %d windows / node digest %s on the whole workload, the GPU %d / %d / %s: no line printed"
                                 % (cpu["nodes"], cpu["windows"], cpu["node_digest"], graph["nodes"], graph["windows"], node_digest))
        want = expected_graph(args, world, shard_reads, total_bases)
        if args.multik:
            graph["nodes_per_k"] = [list(x) for x in per_k]
        if want is not None and not os.environ.get("MDBG_STOP_PHASE") and any(graph[f] != want[f] for f in want):
            raise SystemExit("bench.py: the graph of this run %r differs from the recorded one %r (tests/golden/bench_counts.json): no line printed" % (graph, want))
        if human and args.multik:
            wl = ("multik sweep k = %s on synthetic human %.0f Mb @%.0fx (BASELINE.json configs[4]): %.1f Gbases, l=%d d=%g, sketched ONCE per step, then one graph per k from the "
                  "resident sketches; value = bases / time of the whole sweep, multik_graph_gbases_per_s = bases x %d / time" % (",".join(map(str, MULTIK)), args.genome_mb, args.coverage, total_bases / 1e9, args.l, args.density, len(MULTIK)))
        elif human:
            wl = ("synthetic human %.0f Mb @%.0fx (BASELINE.json configs[3]): %.1f Gbases of ~15 kb HiFi-shaped reads, 0.1%% errors, held as %d shards; the SAME data set at "
                  "every N: %d batch(es) per rank and step" % (args.genome_mb, args.coverage, total_bases / 1e9, HUMAN_SHARDS, len(batches)))
        else:
            wl = "synthetic D. melanogaster %.0f Mb @%.0fx per GPU (BASELINE.json configs[2]): ~15 kb HiFi-shaped reads, 0.1%% errors" % (args.genome_mb, args.coverage)
        if not routed:
            par = "single GPU"
        else:
            if cdist is not None:
                how = "%s exchanged by grouped ncclSend/ncclRecv inside libmdbg_hip.so (mdbg_dist.h) in %d chunks per batch overlapping the tile kernel" % (
                    "window lists + the sketch hashes they need" if args.dist_exchange == "segments" else "whole sketches + window lists", n_chunks)
                if host_comm:
                    how = how.replace("by grouped ncclSend/ncclRecv inside libmdbg_hip.so (mdbg_dist.h)", "by the same layer (mdbg_dist.h) through HOST memory over gloo")
            else:
                how = ("sketches exchanged by send/recv pairs in %d chunks overlapping the tile kernel" % n_chunks) if replicate else "all-to-all of k-min-mer records"
            par = "reads sharded by record x%d, table partitioned by key; %s%s" % (world, how, "" if host_comm else " over RCCL")
            if host_comm:
                par += "; DRY RUN (--comm host): %d ranks on %d GPU(s), not RCCL, not a result" % (world, min(world, n_dev))
        out = {"metric": "Gbases/s ingested to k-min-mer graph", "value": value, "unit": "Gbases/s", "n_gpus": n_ranks, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if human else "weak", "vs_baseline": None,
               "dtype": "u64", "data": "synthetic",
               "config": {"workload": wl, "workload_key": args.workload + ("-multik" if args.multik else ""), "multik": MULTIK if args.multik else None, "k": args.k, "l": args.l, "density": args.density, "minabund": args.minabund,
                          "genome_mb": args.genome_mb, "coverage": args.coverage, "reads_per_gpu": reads_per_gpu,
                          "bases_per_gpu": n_bases, "batches_per_step": len(batches), "total_bases": total_bases, "input_format": args.input, "plain": bool(args.plain),
                          "input": "2-bit packed (two 32-bit planes per 32 bases) resident in HBM" if packed else "ASCII resident in HBM", "parallelism": par,
                          "comm": None if not routed else ("host-staged over gloo: DRY RUN, not RCCL" if host_comm else "rccl")},
               "roofline": roof, "roofline_ascii": roof_ascii, "roofline_hpc_input": roof_hpc, "syncmers": sync_leg, "ascii_in": ascii_in,
               "value_ascii_in": ascii_in["value"] if ascii_in else None, "ms_per_step_ascii_in": ascii_in["ms_per_step"] if ascii_in else None, "pack_ms": pack_ms,
               "cpu_baseline": cpu, "multik_graph_gbases_per_s": multik_graph_rate, "multik_graphs_per_s": (len(MULTIK) * args.steps / dt) if args.multik else None,
               "stage_ms_last_step": {"sketch": st["ms_sketch"], "sketch_bs_kernel": st["ms_sketch_tile"], "insert": st["ms_insert"], "finalize": st["ms_finalize"],
                                      "measured_in": "one further step after the timed region with the stage events on (mdbg_set_timing(2)); the timed steps record the tile kernel's events only"
                                                     if timed_level == 1 else "the last timed step (stage events on in the timed region: MDBG_BENCH_STAGE_EVENTS)"},
               "graph": {"minimizers": st["n_minimizers"], "windows": st["n_windows"], "distinct": st["n_distinct"], "nodes": int(n_nodes), "node_digest": node_digest,
                         "checked_against_recorded_counts": want is not None, "digest_checked_against_recorded": bool(want is not None and "node_digest" in want),
                         "slow_tiles": st["n_slow_tiles"], "tiles": st["n_tiles"], "table_capacity": st["table_capacity"],
                         "partitions_add_up": consistent, "nodes_per_k": per_k if args.multik else None},
               "exchange": exchange, "no_exchange_anchor": anchor, "n1_same_workload": n1_same, "scale_anchor_n1": anchor1,
               "edges_after_timed_region": edges, "side_errors": side_errors or None}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    m.close()
    if cdist is not None:
        cdist.close()
    if routed and cdist is None and engine.tm is not m:
        engine.tm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
