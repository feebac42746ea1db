#!/bin/bash
# round 5, GPU run A: vertical-filter prototype timings, VALU issue rate vs waves per SIMD, what SQ_THREAD_CYCLES_VALU counts, baseline bench + per-phase SQ counters
set -u
R=$(pwd); O=$R/gpurun_out/r5a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for b in vt_proto_l12_u1 vt_proto_l12_u4 vt_proto_l12_u32 vt_proto_l14_u4; do echo "== $b"; timeout 60 $R/scratch/vt_proto/$b 8192 3; done
} > $O/vt_proto.txt 2>&1
{
for W in 1 2 3 4 6 8; do echo "== WGS=$W (waves per SIMD), 8 independent chains per thread"; WGS=$W ONLY="v_xor_b32|v_bitop3_b32|v_alignbit_b32|v_perm_b32" timeout 60 $R/scratch/ubench/valu_rate; done
} > $O/valu_rate_occupancy.txt 2>&1
# which unit does SQ_THREAD_CYCLES_VALU count in: full-rate vs half-rate instruction streams of the same length
ONLY="v_xor_b32|v_alignbit_b32" timeout 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/tc -o q -- $R/scratch/ubench/valu_rate > $O/tc.log 2>&1
python - $O/tc/q_counter_collection.csv > $O/thread_cycles_probe.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])): agg[r["Kernel_Name"] + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items(): print(k, {a: "%.5g" % b for a, b in sorted(v.items())})
PY
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
for P in 1 2 3 0; do
  MDBG_STOP_PHASE=$P timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/ph$P -o q -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --plain > $O/ph$P.log 2>&1
  python - $O/ph$P/q_counter_collection.csv $O/ph$P/q_kernel_trace.csv $P >> $O/phase_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_bs" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[2])) if "sketch_bs" in r["Kernel_Name"]]
print("stop_after", sys.argv[3], "launches", len(dur), "ms", ["%.3f" % d for d in dur], {k: "%.6g" % v for k, v in sorted(agg.items())})
PY
done
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
ls -la $O
