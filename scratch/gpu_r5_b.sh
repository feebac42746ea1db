#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/scratch/vt_proto/vt_rate > $O/vt_rate.txt 2>&1
{ for b in vt_y_n2_w3_u4 vt_y_n4_w4_u1 vt_y_n2_w4_u4 vt_y_n4_w5_u4; do echo "== $b"; timeout 60 $R/scratch/vt_proto/$b 8192 3; done; } > $O/vt_y.txt 2>&1
