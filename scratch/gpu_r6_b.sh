#!/bin/bash
# round 6, second call: the new GPU tests (digest, wrap between two finalize calls, bench refusing a corrupted table), the default line with the digest, the human digest
set -u
R=$(pwd); O=$R/gpurun_out/r6b; mkdir -p $O
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_bench_dry_run.py -x -q 2>&1 | tail -15
python bench.py > $O/default.json 2> $O/default.err; tail -3 $O/default.err
python bench.py --gpus 1 --workload human --cpu-seconds 0 > $O/human_n1.json 2> $O/human_n1.err
python - <<PY
import json
for f in ('default', 'human_n1'):
    try:
        j = json.load(open('$O/%s.json' % f)); print(f, round(j['value'], 1), round(j['ms_per_step'], 3), j['stage_ms_last_step'], j['graph'].get('node_digest'), (j.get('cpu_baseline') or {}).get('node_digest'), (j.get('cpu_baseline') or {}).get('matches_gpu'))
        if j.get('roofline_hpc_input'): print('  hpc leg', {k: j['roofline_hpc_input'][k] for k in ('avg_launch_ms', 'frac', 'minimizers_per_base', 'vs_timed_kernel')})
        if j.get('scale_anchor_n1'): print('  anchor', j['scale_anchor_n1'].get('value'), j['scale_anchor_n1'].get('graph'))
    except Exception as e: print(f, 'failed', e)
PY
