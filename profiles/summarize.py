#!/usr/bin/env python3
"""Condenses a rocprofv3 output set (gpurun_out/pN/{stats,fetch,write}) into the summaries committed under profiles/.

  stats : rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0
  fetch : rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   -- python bench.py --steps 1 --warmup 1 --cpu-seconds 0
  write : rocprofv3 --kernel-trace --pmc WRITE_SIZE ...   (separate pass: FETCH_SIZE takes 3 and WRITE_SIZE 2 of the 4 TCC slots)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section).
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
    json.dump({"note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, see profiles/summarize.py", "kernels": pm}, open(out + "_pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
