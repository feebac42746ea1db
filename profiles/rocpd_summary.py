#!/usr/bin/env python3
"""Turns a rocprofv3 (--kernel-trace --stats) rocpd database into the per-kernel summary kept under profiles/."""
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        f.write("%-70s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in rows:
            f.write("%-70s %8d %14.1f %12.2f %6.2f%%\n" % (name.split("(")[0][:70], calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
