"""one-off: many more seeds of the randomized parity sweep (tests/test_gpu_fuzz.py)"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_fuzz as F
import test_gpu_edges as E
t = time.time(); n = 0
for seed in range(1060, 1060 + int(sys.argv[1])):
    F.test_fuzz_sketch_and_nodes.__wrapped__(seed) if hasattr(F.test_fuzz_sketch_and_nodes, "__wrapped__") else F.test_fuzz_sketch_and_nodes(seed)
    n += 1
for seed in range(100, 100 + int(sys.argv[1]) // 10):
    for ps in (0.0, 0.01, 0.3):
        E.test_gpu_edges_equal_host_emitter_and_oracle(seed, ps)
print("ok", n, "node cases,", int(sys.argv[1]) // 10 * 3, "edge cases in %.1f s" % (time.time() - t))
