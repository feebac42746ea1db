#!/bin/bash
# round 4: where does a multi-GB (re)allocation spend its time — alone, and as test ~500 of the whole pytest session
O=gpurun_out/r4c; mkdir -p $O
MDBG_ALLOC_TRACE=$O/alloc_alone.txt timeout 600 python -m pytest tests/test_gpu_config34.py -x -q -m gpu -k whole_genome_streamed 2>&1 | tail -3 > $O/alone.txt
cp gpurun_out/full_human.json $O/full_human_alone.json
MDBG_ALLOC_TRACE=$O/alloc_session.txt timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/session.txt
cp gpurun_out/full_human.json $O/full_human_session.json
cat $O/alone.txt $O/session.txt
python - <<'PY'
import json
for n in ("alone", "session"):
    j = json.load(open("gpurun_out/r4c/full_human_%s.json" % n))
    print(n, [round(b["ingest_ms"], 1) for b in j["batches"]], {k: v for k, v in j.items() if k not in ("batches", "config")})
PY
grep -c . $O/alloc_alone.txt $O/alloc_session.txt
awk '{ for (i = 1; i <= NF; i++) if ($i ~ /^(malloc|copy|free)=/) { split($i, a, "="); if (a[2] + 0 > 20) { print; break } } }' $O/alloc_session.txt | tail -30
