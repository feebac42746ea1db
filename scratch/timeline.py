"""kernel timeline of the last step of a bench.py run traced with `rocprofv3 --kernel-trace --output-format csv` (scratch/gpu_r6_*.sh).
MODE=def: one step period of the default workload (tile launch to tile launch), every kernel with the gap in front of it.
MODE=human: the last step of the human workload (8 or 16 tile launches), gaps, time per kernel name, and the kernels from the last insertion on (= finalize)."""
import collections
import csv
import os
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
sk = [i for i, r in enumerate(rows) if "sketch_bs_kernel" in r[2]]
if os.environ.get("MODE", "def") == "def":
    a, b = sk[-2], sk[-1]
    t0 = rows[a][0]
    print("step period %.3f ms (tile launch to tile launch), %d launches" % ((rows[b][0] - t0) / 1e6, b - a))
    prev = rows[a][0]; busy = 0
    for s, e, n in rows[a:b]:
        print("%8.3f ms  gap %7.3f  dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6 if s > prev else 0.0, (e - s) / 1e6, n))
        busy += max(0, e - max(s, prev)); prev = max(prev, e)
    print("busy %.3f ms, idle %.3f ms" % (busy / 1e6, (rows[b][0] - t0 - busy) / 1e6))
elif os.environ.get("MODE") == "dist":
    # the last step that went through the multi-GPU layer: from the fin_emit in front of the last owner_list_count_kernel to the fin_emit behind it
    ol = [i for i, r in enumerate(rows) if r[2].startswith("owner_list_count")]
    fe = [i for i, r in enumerate(rows) if r[2].startswith("fin_emit")]
    a = max([i for i in fe if i < ol[-1]] + [-1]) + 1
    b = min([i for i in fe if i > ol[-1]])
    seg = rows[a:b + 1]
    t0 = seg[0][0]; busy = 0; gaps = []; prev_end = seg[0][0]
    for s, e, n in seg:
        if s > prev_end: gaps.append((s - prev_end, n, (s - t0) / 1e6))
        busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
    span = prev_end - t0
    print("last dist step: span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps, %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), len(seg)))
    for g, n, at in sorted(gaps, reverse=True)[:30]: print("gap %.3f ms before %s at %.2f ms" % (g / 1e6, n, at))
    c = collections.Counter(); cn = collections.Counter()
    for s, e, n in seg: c[n] += e - s; cn[n] += 1
    for n, t in c.most_common(50): print("%-50s %4d x  %.3f ms" % (n, cn[n], t / 1e6))
    ins = [i for i, r in enumerate(seg) if r[2].startswith("insert_")]
    print("from the last insertion on (start, gap, dur in ms):")
    prev = seg[ins[-1]][0]
    for s, e, n in seg[ins[-1]:]:
        print("  %8.3f gap %7.3f dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6 if s > prev else 0.0, (e - s) / 1e6, n)); prev = max(prev, e)
    print("the first chunk (start, gap, dur in ms):")
    prev = seg[0][0]
    for s, e, n in seg[:60]:
        print("  %8.3f gap %7.3f dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6 if s > prev else 0.0, (e - s) / 1e6, n)); prev = max(prev, e)
else:
    nt = int(os.environ.get("TILES_PER_STEP", "0")) or (16 if len(sk) % 16 == 0 and len(sk) >= 32 else 8)
    i0 = sk[-nt]
    while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 2_000_000 and not rows[i0 - 1][2].startswith("fin_emit"): i0 -= 1
    seg = rows[i0:]
    t0 = seg[0][0]; busy = 0; gaps = []; prev_end = seg[0][0]
    for s, e, n in seg:
        if s > prev_end: gaps.append((s - prev_end, n, (s - t0) / 1e6))
        busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
    span = prev_end - t0
    print("last step: span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps, %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), len(seg)))
    for g, n, at in sorted(gaps, reverse=True)[:20]: print("gap %.3f ms before %s at %.2f ms" % (g / 1e6, n, at))
    c = collections.Counter(); cn = collections.Counter()
    for s, e, n in seg: c[n] += e - s; cn[n] += 1
    for n, t in c.most_common(40): print("%-50s %4d x  %.3f ms" % (n, cn[n], t / 1e6))
    ins = [i for i, r in enumerate(seg) if r[2].startswith("insert_")]
    print("from the last insertion on (start, gap, dur in ms):")
    prev = seg[ins[-1]][0]
    for s, e, n in seg[ins[-1]:]:
        print("  %8.3f gap %7.3f dur %7.3f  %s" % ((s - t0) / 1e6, (s - prev) / 1e6 if s > prev else 0.0, (e - s) / 1e6, n)); prev = max(prev, e)
