#!/usr/bin/env python3
"""Condenses a rocprofv3 output set (gpurun_out/pN/{stats,fetch,write}) into the summaries committed under profiles/.

  stats : rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0
  fetch : rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   -- python bench.py --steps 1 --warmup 1 --cpu-seconds 0
  write : rocprofv3 --kernel-trace --pmc WRITE_SIZE ...   (separate pass: FETCH_SIZE takes 3 and WRITE_SIZE 2 of the 4 TCC slots)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section).
  sq, sq2 : two passes of 8 SQ counters each (scratch/gpu_profile_set.sh); the launches of the bench's ASCII side measurement
            (roofline_ascii) are told apart from the timed packed ones by their order: the bench runs them last.
usage: summarize.py <dir> <out_prefix>
"""
import csv
import json
import sys
from collections import defaultdict


def counters(path, name):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main(d, out):
    with open(out + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0   (durations in microseconds)\n")
        f.write("%-48s %6s %12s %12s %12s %7s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in csv.DictReader(open(d + "/stats/s_kernel_stats.csv")):
            f.write("%-48s %6s %12.1f %12.1f %12.1f %6s%%\n" % (r["Name"].split("(")[0].replace("void ", "")[:48], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    fe, wr = counters(d + "/fetch/f_counter_collection.csv", "FETCH_SIZE"), counters(d + "/write/w_counter_collection.csv", "WRITE_SIZE")
    pm = {}
    for k in fe:
        f_kb = fe[k][1] / fe[k][0]
        w_kb = wr[k][1] / wr[k][0] if k in wr and wr[k][0] else 0.0
        pm[k] = {"launches_sampled": fe[k][0], "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
                 "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}
    json.dump({"note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, see profiles/summarize.py; averaged over every launch of the run "
                       "(with --input packed the bench also launches the tile kernel twice on ASCII input for roofline_ascii, see per_launch)",
               "kernels": pm, "per_launch": per_launch(d), "tile_kernel_by_input": by_input(d)}, open(out + "_pmc_traffic.json", "w"), indent=1)
    sq_summary(d, out)


def per_launch(d):
    """FETCH/WRITE of every single launch of the tile kernel, in launch order"""
    out = {}
    for sub, f, name in (("fetch", "f", "FETCH_SIZE"), ("write", "w", "WRITE_SIZE")):
        try:
            rows = [r for r in csv.DictReader(open("%s/%s/%s_counter_collection.csv" % (d, sub, f))) if r["Counter_Name"] == name and "sketch_bs_kernel" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            out[name + "_KB"] = [float(r["Counter_Value"]) for r in rows]
        except Exception as e:      # noqa: BLE001
            out[name + "_KB"] = str(e)
    return out


def by_input(d):
    """HBM bytes per launch of the tile kernel by input format: with --input packed a bench run launches the tile kernel
    warmup + steps times on packed input, twice on ASCII (roofline_ascii) and once more on packed input (the table the edge
    stage refers to)"""
    try:
        fmt = json.load(open(d + "/bench.json"))["config"]["input_format"]
    except Exception:      # noqa: BLE001
        return None
    pl = per_launch(d)
    f, w = pl.get("FETCH_SIZE_KB"), pl.get("WRITE_SIZE_KB")
    if not isinstance(f, list) or not isinstance(w, list) or len(f) != len(w) or not f:
        return None
    hbm = [(2.0 * a + b) * 1024.0 for a, b in zip(f, w)]
    if json.load(open(d + "/bench.json"))["config"].get("plain"):      # round 4: profiler runs are --plain (every launch is of the timed kind); the ASCII side comes from its own set
        return {fmt: sum(hbm) / len(hbm)}
    if fmt != "packed" or len(hbm) < 4:
        return {fmt: sum(hbm) / len(hbm)}
    pk = hbm[:-3] + hbm[-1:]
    return {"packed": sum(pk) / len(pk), "ascii": sum(hbm[-3:-1]) / 2.0}


def sq_summary(d, out):
    import os
    try:
        bench = json.load(open(d + "/bench.json"))
    except Exception:      # noqa: BLE001
        return
    agg = defaultdict(lambda: defaultdict(list))
    for sub in ("sq", "sq2"):
        p = "%s/%s/q_counter_collection.csv" % (d, sub)
        if not os.path.exists(p):
            return
        rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    cfg = bench["config"]
    tile = [k for k in agg if k.startswith("sketch_bs_kernel")]
    res = {"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python bench.py --steps 1 --warmup 1 --cpu-seconds 0; per launch; "
                   "bench.py --plain: every launch is one of the timed kind (warm-up + steps)",
           "config": {f: cfg[f] for f in ("k", "l", "density", "minabund", "input_format", "bases_per_gpu")}, "kernels": {}}
    for k in agg:
        if not (k.startswith("sketch_bs") or k.startswith("insert_windows") or k.startswith("fin_") or k.startswith("gather")):
            continue
        res["kernels"][k] = {c: v for c, v in agg[k].items()}
    if tile:
        c = agg[tile[0]]
        first = lambda name: c[name][0]
        nb = cfg["bases_per_gpu"]
        res["derived"] = {
            "valu_lane_ops_per_base": first("SQ_INSTS_VALU") * 64.0 / nb,
            "valu_instr_per_simd_cycle": first("SQ_INSTS_VALU") / (first("SQ_BUSY_CU_CYCLES") * 4.0),
            # (rounds 2-3 also derived a "valu_util" from SQ_ACTIVE_INST_VALU, which counts a quad-cycle per instruction whatever it costs and so read 1.03: dropped.
            #  VALU cycles estimated from the instruction mix and the measured issue rates, profiles/r01_g_valu_rates.txt: ~77 % of the kernel's SIMD cycles)
            "wave_time_split": {"active": first("SQ_ACTIVE_INST_ANY") / first("SQ_WAVE_CYCLES"), "parked_waitcnt_barrier": first("SQ_WAIT_ANY") / first("SQ_WAVE_CYCLES"),
                                "issue_stall": first("SQ_WAIT_INST_ANY") / first("SQ_WAVE_CYCLES")},
            "salu_per_valu": first("SQ_INSTS_SALU") / first("SQ_INSTS_VALU"), "lds_per_valu": first("SQ_INSTS_LDS") / first("SQ_INSTS_VALU")}
    json.dump(res, open(out + "_sq_counters.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
