"""bench.py's N>1 path executed for real before an 8-GPU node ever sees it: two OS processes, the library's multi-GPU layer (include/mdbg_dist.h),
the exchange staged through host memory over gloo (--comm host) because two ranks share this box's one GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _bench(*flags, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_dry_run_matches_one_rank():
    small = ["--workload", "human", "--genome-mb", "40", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    one = _bench("--gpus", "1", *small)
    two = _bench("--gpus", "2", "--comm", "host", *small)
    assert one["n_gpus"] == 1 and one["config"]["batches_per_step"] == 8 and one["scaling"] == "strong"
    assert two["n_gpus"] == 2 and two["config"]["batches_per_step"] == 4 and two["scaling"] == "strong"
    assert two["config"]["total_bases"] == one["config"]["total_bases"] and 2 * two["config"]["bases_per_gpu"] != 0
    # the same data set, the same graph: the partitions add up to the node count one context finds
    assert two["graph"]["partitions_add_up"] is True and two["graph"]["nodes"] == one["graph"]["nodes"] > 1000
    # ... and the same SET of (key, abundance) pairs: the digests of the two partitions add / XOR up to the one-GPU table's (round 6)
    assert one["graph"]["node_digest"] and two["graph"]["node_digest"] == one["graph"]["node_digest"]
    assert two["exchange"]["bytes_in_busiest_rank_per_step"] > 0 and "not RCCL" in two["exchange"]["transport"]
    assert two["no_exchange_anchor"] and two["no_exchange_anchor"]["value"] > 0
    assert "DRY RUN" in two["config"]["parallelism"] and "not RCCL" in two["config"]["comm"]
    assert two["roofline"]["kernel"] == "sketch_bs_kernel<14>" and one["roofline"]["launches_per_step"] == 8


@pytest.mark.gpu
def test_bench_default_line_carries_the_ascii_leg():
    j = _bench("--gpus", "1", "--genome-mb", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0")
    a = j["ascii_in"]
    assert j["config"]["workload_key"] == "fly" and a["ms_per_step"] > 0 and a["pack_ms"] > 0 and 0 < a["value_pack_then_packed"] < j["value"]
    h = j["roofline_hpc_input"]          # round 6: the same launch with reads_already_hpc = 1 (README.md:134: the condition of the published timings)
    assert h["avg_launch_ms"] > 0 and h["minimizers_per_base"] > j["graph"]["minimizers"] / j["config"]["bases_per_gpu"] and 0 < h["frac"] < 1
    assert "valu_util" not in j["roofline"]
    sy = j["syncmers"]                   # row f4: the same reads under --syncmers -s 4 (the tile kernel's syncmer instantiation)
    assert sy["l"] == 12 and sy["s"] == 4 and sy["kernel_gbases_per_s"] > 0 and sy["sketch_ms"] >= sy["kernel_ms"] > 0 and 0.001 < sy["minimizers_per_base"] < 0.02


@pytest.mark.gpu
def test_bench_cpu_leg_checks_the_whole_workload_against_the_oracle():
    """when the CPU sample is the whole workload (a small genome here; configs[2] in the driver's run) the line carries the oracle's node and
    window counts of that very workload and has been refused unless the GPU's are the same (src/main.rs:926-932 prints the same counters)"""
    j = _bench("--gpus", "1", "--genome-mb", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "60", "--no-scale-anchor")
    c = j["cpu_baseline"]
    assert c["whole_workload"] is True and c["matches_gpu"] is True
    assert c["nodes"] == j["graph"]["nodes"] > 1000 and c["windows"] == j["graph"]["windows"] > 1000
    # round 6: not two counts but the node SET — the oracle's order-free digest over its (key, abundance) pairs equals the digest of the device table
    assert c["node_digest"] == j["graph"]["node_digest"] and len(c["node_digest"]) == 2 and c["node_digest"][0] != "0x%016x" % 0


@pytest.mark.gpu
def test_bench_refuses_the_line_when_one_abundance_differs():
    """the same run with ONE abundance of the device node table changed behind the timed region (a test hook of bench.py): node and window counts are what they were,
    the digest is not — no line, a non-zero exit and the reason on stderr"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MDBG_BENCH_CORRUPT"] = "abundance"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--genome-mb", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "60", "--no-scale-anchor"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.strip()], (r.returncode, r.stdout[-500:])
    assert "no line printed" in r.stderr and "node digest" in r.stderr, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_multik_sweep_one_rank_and_two_ranks_agree():
    """configs[4]: sketched once, one graph per k from the resident sketches; at N>1 the ranks keep whole sketches, so no k needs a new exchange"""
    small = ["--workload", "human", "--multik", "--genome-mb", "40", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0"]
    one = _bench("--gpus", "1", *small)
    two = _bench("--gpus", "2", "--comm", "host", *small)
    ks = [10, 15, 20, 25, 30, 35, 40]
    assert [k for k, _ in one["graph"]["nodes_per_k"]] == ks and one["config"]["l"] == 12 and one["config"]["multik"] == ks
    assert one["graph"]["nodes_per_k"] == two["graph"]["nodes_per_k"] and all(n > 500 for _, n in one["graph"]["nodes_per_k"])
    assert two["exchange"]["mode"] == "whole" and two["graph"]["partitions_add_up"] is True
    assert one["roofline"]["launches_per_step"] == 8          # one sketch per batch and sweep: nothing is sketched again for the later k
    # the same sweep under the DEFAULT exchange (segments): mdbg_dist_reset(k) exchanges the rounds again for every k (round 5; until then MDBG_E_STATE) — same graphs,
    # still one sketch per batch, and the line says what the seven exchanges moved
    seg = _bench("--gpus", "4", "--comm", "host", "--multik-exchange", "segments", *small)
    assert seg["graph"]["nodes_per_k"] == one["graph"]["nodes_per_k"] and seg["exchange"]["mode"] == "segments" and seg["graph"]["partitions_add_up"] is True
    assert seg["config"]["batches_per_step"] == 2 and seg["roofline"]["launches_per_step"] == 4          # two batches x two chunks per rank, sketched once per sweep
    assert seg["exchange"]["bytes_in_busiest_rank_per_step"] > 0


@pytest.mark.gpu
def test_bench_one_rank_through_the_rccl_transport():
    """the nearest this pool gets to the driver's N>1 run: torch's nccl process group, the library's RCCL communicator (one rank), the multi-GPU layer, the human workload"""
    small = ["--workload", "human", "--genome-mb", "40", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    one = _bench("--gpus", "1", *small)
    rccl = _bench("--gpus", "1", "--force-dist", *small)
    assert rccl["config"]["comm"] == "rccl" and "RCCL" in rccl["exchange"]["transport"] and rccl["graph"]["partitions_add_up"] is True
    assert rccl["graph"]["nodes"] == one["graph"]["nodes"] and rccl["config"]["batches_per_step"] == 8 and rccl["no_exchange_anchor"]["value"] > 0
    assert rccl["graph"]["node_digest"] == one["graph"]["node_digest"]


@pytest.mark.gpu
def test_bench_prints_its_line_when_side_measurements_fail():
    """everything measured beside the headline (recorded profiles, the ASCII leg, the edge stage, the CPU leg, the anchors) may fail: the line is still
    printed, with the failure named under side_errors"""
    every = "pmc_traffic,issue_roofline,sq_counters,ascii_in,roofline_hpc_input,syncmers,edges_after_timed_region,cpu_baseline,n1_same_workload,no_exchange_anchor"
    j = _bench("--gpus", "1", "--genome-mb", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "5", "--no-scale-anchor", MDBG_BENCH_FAIL_SIDE=every)
    assert j["value"] > 0 and j["graph"]["nodes"] > 1000 and j["roofline"]["frac"] > 0
    assert set(j["side_errors"]) == {"pmc_traffic", "issue_roofline", "sq_counters", "ascii_in", "roofline_hpc_input", "syncmers", "edges_after_timed_region", "cpu_baseline"}
    assert j["roofline_hpc_input"] is None and j["syncmers"] is None and j["ascii_in"] is None and j["edges_after_timed_region"] is None and j["cpu_baseline"] is None and j["roofline"]["traffic"] is None
    small = ["--workload", "human", "--genome-mb", "40", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0"]
    two = _bench("--gpus", "2", "--comm", "host", *small, MDBG_BENCH_FAIL_SIDE=every)
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["graph"]["partitions_add_up"] is True
    assert two["no_exchange_anchor"] is None and two["n1_same_workload"] is None and "n1_same_workload" in two["side_errors"]


@pytest.mark.gpu
def test_bench_eight_ranks_dry_run_matches_one_rank():
    """the N = 8 branch the driver's scaling run takes — eight OS processes, one shard each, thin window lists, two rounds per step — with the exchange staged through
    host memory (eight ranks share this box's one GPU): the same node set as one rank, and the line carries the layer's stage timers beside the eight-rank budget"""
    small = ["--workload", "human", "--genome-mb", "160", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0"]
    one = _bench("--gpus", "1", *small)
    eight = _bench("--gpus", "8", "--comm", "host", *small)
    assert eight["n_gpus"] == 8 and eight["config"]["batches_per_step"] == 1 and eight["config"]["total_bases"] == one["config"]["total_bases"]
    assert eight["graph"]["partitions_add_up"] is True and eight["graph"]["nodes"] == one["graph"]["nodes"] > 100000
    assert one["graph"]["node_digest"] and eight["graph"]["node_digest"] == one["graph"]["node_digest"]
    lay = eight["exchange"]["layer_ms_per_step"]
    assert lay["rounds_per_step"] == 2.0 and set(lay["slowest_rank"]) == set(lay["mean"]) and lay["slowest_rank"]["sketch"] > 0 and lay["slowest_rank"]["insert"] > 0
    assert all(lay["slowest_rank"][s_] >= lay["mean"][s_] - 1e-9 for s_ in lay["mean"])
    assert eight["exchange"]["budget_ms_per_step_8_ranks_human"]["sum"][0] > 10 and eight["exchange"]["nodes_busiest_rank_over_mean"] < 1.1
