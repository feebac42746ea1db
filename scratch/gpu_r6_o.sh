#!/bin/bash
# round 6: the syncmer machine as a scan, second version (masks, table indexed by plane bits, prefix / suffix minima); the kernel's time up to the end of phase 2 / 3 / 4
set -u
R=$(pwd); O=$R/gpurun_out/r6o; mkdir -p $O
export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.txt; echo "$(date +%T) $*"; }
say "syncmer tests"; timeout 1500 python -m pytest tests/test_gpu_syncmers.py -x -q > $O/sync_tests.log 2>&1; say "rc $? $(tail -1 $O/sync_tests.log)"
tail -30 $O/sync_tests.log | cut -c1-200
say "measure"; timeout 600 python scratch/measure_syncmers.py > $O/syncmers.json 2> $O/syncmers.err; say "rc $?"; cat $O/syncmers.json
for ph in 2 3 4; do say "stop $ph"; MDBG_STOP_PHASE=$ph timeout 600 python scratch/measure_syncmers.py > $O/syncmers_stop$ph.json 2> /dev/null; python -c "
import json;print([ (r['l'],r['s'],round(r['ms_kernel'],3)) for r in json.load(open('$O/syncmers_stop$ph.json'))])"; done
say done
