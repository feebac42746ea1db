import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_fuzz as F
import test_gpu_edges as E
bad = []; t = time.time()
a, n = int(sys.argv[1]), int(sys.argv[2])
for seed in range(a, a + n):
    try:
        F.test_fuzz_sketch_and_nodes(seed)
    except AssertionError as e:
        bad.append((seed, repr(e)[:60]))
for seed in range(a, a + n // 8):
    for ps in (0.0, 0.01, 0.4):
        try:
            E.test_gpu_edges_equal_host_emitter_and_oracle(seed, ps)
        except AssertionError as e:
            bad.append(("edges", seed, ps, repr(e)[:60]))
print("bad", bad, "in %.0f s" % (time.time() - t))
