"""tile-kernel throughput over the reference's parameter space: l x density, fast path vs generic walker, packed and ASCII input"""
import json, sys
sys.path.insert(0, '.')
import torch
import rust_mdbg_amd as R
n_reads = 133333                     # 2.0 Gbases of the bench reads
rows = []
for l in (12, 20, 31):
    for d in (0.003, 0.03, 0.1):
        m = R.Mdbg(10, l, d, 2)
        db, do, nb = m.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=n_reads)
        words = torch.zeros((nb + 31) // 32 + 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()      # the fills above ran on torch's stream, the packer runs on the context's
        assert m.pack_device(db, nb, words.data_ptr()) == 0
        r = dict(l=l, density=d, gbases=nb / 1e9)
        for name in ("packed", "ascii"):
            for rep in range(2):
                m.reset(0)
                if name == "ascii": m.sketch_device(db, do, n_reads, nb, 0)
                else: m.ingest_packed_device(words.data_ptr(), do, n_reads, nb, 0, sketch_only=True)
            st = m.stats()
            r[name] = dict(ms_tile=st["ms_sketch_tile"], ms_sketch=st["ms_sketch"], tbases_per_s=nb / (st["ms_sketch_tile"] * 1e-3) / 1e12,
                           slow_tiles=st["n_slow_tiles"], tiles=st["n_tiles"], minimizers=st["n_minimizers"], launches=st["n_sketch_tile_launches"])
        m.close()
        g = R.Mdbg(10, l, d, 2, flags=1)           # MDBG_FLAG_FORCE_GENERIC on 1/16 of the reads
        nr = n_reads // 16
        db, do, nb2 = g.synth_reads_device(seed=1, genome_len=140_000_000, n_reads=nr)
        for rep in range(2):
            g.reset(0); g.sketch_device(db, do, nr, nb2, 0)
        st = g.stats()
        r["generic_walker_ascii"] = dict(gbases=nb2 / 1e9, ms_tile=st["ms_sketch_tile"], tbases_per_s=nb2 / (st["ms_sketch_tile"] * 1e-3) / 1e12, slow_tiles=st["n_slow_tiles"], tiles=st["n_tiles"])
        g.close()
        rows.append(r)
        print(json.dumps(r), flush=True)
json.dump(dict(workload="bench reads (synthetic 140 Mb genome, ~15 kb reads), 2.0 Gbases; tile kernel only (HIP events)", rows=rows), open("gpurun_out/param_grid.json", "w"), indent=1)
