import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_fuzz as F
bad = []; t = time.time()
for seed in range(int(sys.argv[1])):
    try:
        F.test_fuzz_long_reads_across_tiles(seed)
    except AssertionError as e:
        bad.append((seed, repr(e)[:80]))
        if len(bad) > 5: break
print("bad", bad, "in %.0f s" % (time.time() - t))
