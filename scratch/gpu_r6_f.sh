#!/bin/bash
# round 6: (1) the multi-GPU layer at one rank after the rows of a partition come out in index order; dist tests; (2) the filter width sweep (variants built with -DMDBG_BS_B=n);
# (3) the gzip reader on this host
set -u
R=$(pwd); O=$R/gpurun_out/r6f; mkdir -p $O
P=$O/progress.txt; : > $P
say() { echo "$(date +%T) $*" >> $P; echo "$(date +%T) $*"; }
for t in tests/test_gpu_dist_c.py tests/test_gpu_distributed.py tests/test_gpu_dist_scale.py tests/test_gpu_round5.py tests/test_gpu_config34.py; do
  say "pytest $t"; timeout 1200 python -m pytest $t -x -q > $O/$(basename $t .py).log 2>&1; say "rc $? $(grep -E 'passed|failed|error' $O/$(basename $t .py).log | tail -1)"
done
say "human dist w1"; MDBG_DIST_TIMING=1 timeout 600 python bench.py --gpus 1 --workload human --force-dist --cpu-seconds 0 --steps 10 > $O/human_dist_w1.json 2> $O/human_dist_w1.err; say "rc $?"
say "human n1"; timeout 600 python bench.py --gpus 1 --workload human --cpu-seconds 0 --steps 10 > $O/human_n1.json 2> $O/human_n1.err; say "rc $?"
python - <<PY
import json
for f in ('human_dist_w1', 'human_n1'):
    try:
        j = json.load(open('$O/%s.json' % f)); s = j['stage_ms_last_step']
        print(f, round(j['value'], 1), round(j['ms_per_step'], 3), {k: round(v, 3) for k, v in s.items() if k != 'measured_in'}, (j.get('no_exchange_anchor') or {}).get('ms_per_step'), j['graph'].get('node_digest'))
    except Exception as e: print(f, 'failed', e)
PY
say "bs_b sweep"
cp rust_mdbg_amd/libmdbg_hip.so $O/keep.so
for rep in 1 2; do for b in 8 6 7 9 10; do
  cp scratch/variants/bsb$b.so rust_mdbg_amd/libmdbg_hip.so
  for cfg in "12 0.002" "14 0.003"; do set -- $cfg
    timeout 300 python bench.py --cpu-seconds 0 --no-scale-anchor --plain -l $1 --density $2 > $O/bsb.json 2> $O/bsb.err
    python - "$b" "$1" "$2" <<PY >> $O/bs_b_sweep.txt
import json, sys
try:
    j = json.load(open('$O/bsb.json')); print('BS_B=%s l=%s d=%s: tile kernel %.4f ms, step %.4f ms, minimizers %d, nodes %d' % (sys.argv[1], sys.argv[2], sys.argv[3], j['roofline']['avg_launch_ms'], j['ms_per_step'], j['graph']['minimizers'], j['graph']['nodes']))
except Exception as e: print('BS_B=%s l=%s d=%s failed: %r' % (sys.argv[1], sys.argv[2], sys.argv[3], e))
PY
  done
done; done
cp $O/keep.so rust_mdbg_amd/libmdbg_hip.so; rm -f $O/keep.so
cat $O/bs_b_sweep.txt
say "gz on this host"; timeout 1500 bash scratch/gpu_r6_gz.sh > $O/gz.txt 2>&1; say "rc $?"; cat $O/gz.txt
say done
