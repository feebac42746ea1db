#!/usr/bin/env python3
"""Bytes a rank receives per step in the sketch-exchange mode (include/mdbg_dist.h) at the BASELINE configs[3] shard, for W = 2 / 4 / 8,
from the counts measured on the full data set (profiles/r03_full_human.json: minimizers, windows and reads of one 19.5-Gbase shard), and
the wire time they imply.  xGMI is point to point: a rank's W-1 peers each sit behind their own link, all links run in parallel, so the
time is set by ONE peer's bytes over ONE link direction.  Link rates: 76.8 GB/s per direction (153.6 GB/s bidirectional, the published
MI355X figure) and 60 GB/s (what RCCL send/recv typically sustains).
usage: exchange_budget.py > profiles/r03_exchange_budget.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fh = json.load(open(os.path.join(ROOT, "profiles", "r03_full_human.json")))
b0 = fh["batches"][0]
n_min, n_win, n_reads, bases = b0["minimizers_total"], b0["windows_total"], fh["config"]["reads"] // fh["config"]["batches"], b0["bases"]
out = {"source": "profiles/r03_full_human.json, first batch", "shard": {"bases": bases, "reads": n_reads, "minimizers": n_min, "window_occurrences": n_win},
       "per_peer_bytes": "8 * minimizers (hashes) + 8 * (reads + 1) (per-read offsets) + 8 * windows / W (the windows that peer owns, (start, read) pairs)",
       "round2_per_peer_bytes": "12 * minimizers + 8 * (reads + 1) + 8 * windows / W  (hashes + positions)", "worlds": []}
for W in (2, 4, 8):
    per_peer = 8 * n_min + 8 * (n_reads + 1) + 8 * n_win // W
    per_peer_r2 = 12 * n_min + 8 * (n_reads + 1) + 8 * n_win // W
    row = {"world": W, "bytes_per_peer": per_peer, "bytes_in_per_rank_per_step": per_peer * (W - 1), "round2_bytes_in_per_rank_per_step": per_peer_r2 * (W - 1),
           "saved": 1 - per_peer / per_peer_r2}
    for name, rate in (("76.8_GBps_per_direction", 76.8e9), ("60_GBps_per_direction", 60e9)):
        row["wire_ms_at_" + name] = per_peer / rate * 1e3
        row["round2_wire_ms_at_" + name] = per_peer_r2 / rate * 1e3
    # the position fetch at finalize: solid nodes this rank owns whose A-th sighting lies in another rank's reads
    nodes = fh["nodes"] / 8 * W / W            # nodes per shard-equivalent of genome (weak scaling: nodes grow with W)
    q = nodes * (W - 1) / W
    row["finalize_position_fetch_bytes_per_rank"] = int(q * (8 + 16) * 2)      # queries out + answers in, and the same served for the peers
    out["worlds"].append(row)
out["compute_ms_per_rank_per_step"] = {"measured": "profiles/r02_notes.md: 13.9 / 15.0 / 15.9 ms for W = 1 / 2 / 4 (emulated, GPU time per rank, round 2's kernels)",
                                       "note": "the exchange runs in 4 chunks on its own stream under the tile kernel of the next chunk and (this round) under the insertion of the previous one"}
# segments (the default exchange from the second half of round 3): measured, not modelled — W thread-ranks of the C layer on one GPU
# (scratch/measure_dist_traffic.py -> profiles/r03_dist_traffic.jsonl), bytes into the busiest rank per step incl. the position fetch at finalize
seg = {"what": "mdbg_dist_set_exchange(MDBG_EXCHANGE_SEGMENTS): window lists + only the hashes the listed windows need; owner = rank of the window's smallest hash",
       "source": "profiles/r03_dist_traffic.jsonl (measured through mdbg_dist_traffic; config 2 = 7.0 Gbases per rank l=12 d=0.002, config 3 = 19.5 Gbases per rank l=14 d=0.003)", "rows": []}
for line in open(os.path.join(ROOT, "profiles", "r03_dist_traffic.jsonl")):
    j = json.loads(line); W = j["world"]
    row = {"world": W, "config": j["config"]}
    for mode in ("segments", "whole"):
        m = j[mode]
        row[mode] = {"bytes_in_busiest_rank": m["bytes_in_per_rank_per_step_max"], "bytes_in_mean": m["bytes_in_per_rank_per_step_mean"],
                     "per_peer_wire_ms_at_76.8_GBps": m["bytes_in_per_rank_per_step_max"] / (W - 1) / 76.8e9 * 1e3, "per_peer_wire_ms_at_60_GBps": m["bytes_in_per_rank_per_step_max"] / (W - 1) / 60e9 * 1e3}
    nl = j["segments"]["nodes_local"]
    row["nodes_per_rank_max_over_mean"] = max(nl) / (sum(nl) / W)
    seg["rows"].append(row)
s3 = [r for r in seg["rows"] if r["config"] == 3]
if s3:
    r = s3[-1]; W = r["world"]; per_rank_out = r["segments"]["bytes_in_mean"] * W / (W - 1)
    seg["extrapolated_config3_world8"] = {"note": "the segment volume a rank SENDS does not depend on W (its windows' hashes go out once, to whoever owns them); a rank receives (W-1)/W of the mean",
                                           "bytes_in_mean": per_rank_out * 7 / 8, "per_peer_wire_ms_at_76.8_GBps": per_rank_out / 8 / 76.8e9 * 1e3, "per_peer_wire_ms_at_60_GBps": per_rank_out / 8 / 60e9 * 1e3}
out["segments_measured"] = seg
json.dump(out, sys.stdout, indent=1)
print()
