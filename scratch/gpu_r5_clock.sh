#!/bin/bash
# what does the shader clock read WHILE the tile kernel runs back to back?  (the issue roofline prices the SIMDs' cycles at 2.4 GHz)
cd /root/repo; O=gpurun_out/r5clk; mkdir -p $O; rm -f $O/started $O/busy.txt
python - > $O/tile.txt 2>&1 <<'PY' &
import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import rust_mdbg_amd as R
m = R.Mdbg(35, 12, 0.002, 2, device=0)
n_reads = 466666
db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert m.pack_device(db, nb, words.data_ptr()) == 0
for it in range(20):
    m.reset(0); m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
m.sync()
open("gpurun_out/r5clk/started", "w").write("x")
t = time.time(); tot = 0.0; n = 0
while time.time() - t < 9.0:
    m.reset(0); m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
    tot += m.stats()["ms_sketch_tile"]; n += 1
print("tile kernel %.4f ms (mean of %d back-to-back sketch-only steps, %.1f s)" % (tot / n, n, time.time() - t))
PY
P=$!
for i in $(seq 1 600); do [ -f $O/started ] && break; sleep 0.5; done
for i in $(seq 1 12); do
  (echo "== sample $i"; rocm-smi --showmetrics 2>&1 | grep -i -E "current_gfxclk|current_socket_power|current_uclk|throttle" | head -8) >> $O/busy.txt
  sleep 0.2
done
wait $P
cat $O/tile.txt; grep -i -E "sample|gfxclk|power|uclk" $O/busy.txt | head -60
