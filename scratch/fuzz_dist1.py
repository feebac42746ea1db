import sys, random, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_distributed as D
import test_gpu_fuzz as F
seed = int(sys.argv[1])
rnd = random.Random(9000 + seed)
k, l, d, a = rnd.choice([(2, 8, 0.03, 1), (3, 8, 0.05, 2), (5, 10, 0.01, 2), (7, 12, 0.008, 3), (21, 12, 0.004, 2), (4, 6, 0.05, 8)])
reads = F.fuzz_reads(rnd, n_reads=rnd.randint(3, 150), genome_len=rnd.choice([300, 5000, 40000]), mean_len=rnd.choice([40, 400, 4000]),
                     err=rnd.choice([0.0, 0.02]), p_lower=0.0, p_n=rnd.choice([0.0, 0.2]), p_hp=rnd.choice([0.0, 0.02]))
reads = [r.replace(b"n", b"N") for r in reads]
world = rnd.choice([1, 2, 3, 5]); mode = rnd.choice(["route", "replicate", "replicate-pipelined", "replicate-pipelined-nosize"])
bpr = rnd.choice([1, 2, 3])
print(seed, world, mode, (k, l, d, a), len(reads), [len(r) for r in reads][:30], bpr)
try:
    D.run(world, reads, k, l, d, a, batches_per_rank=bpr, mode=mode)
except AssertionError as e:
    for x in e.args[0] if isinstance(e.args[0], list) else [e]:
        print("ERR:", repr(x)[:300])
        traceback.print_exception(type(x), x, x.__traceback__, limit=6)
