#!/bin/bash
# PC sampling probe (beta): does the box support it, and what does a sample carry?
set -u
R=$(pwd); O=$R/gpurun_out/r5c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d $O/st -o q -- python $R/bench.py --steps 2 --warmup 0 --cpu-seconds 0 --plain > $O/st.log 2>&1
echo "stochastic rc=$?" >> $O/st.log
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d $O/ht -o q -- python $R/bench.py --steps 2 --warmup 0 --cpu-seconds 0 --plain > $O/ht.log 2>&1
echo "host_trap rc=$?" >> $O/ht.log
for d in st ht; do ls -la $O/$d 2>/dev/null; for f in $O/$d/*pc_sampling*.csv; do [ -f "$f" ] && { wc -l $f; head -3 $f; }; done; done > $O/summary.txt 2>&1
for f in $O/*/*pc_sampling*.csv; do [ -f "$f" ] && head -c 40000000 $f > $f.cut && rm $f; done
find $O -name "*.db" -delete
