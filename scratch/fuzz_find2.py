import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_fuzz as F
bad = []
for seed in [1230, 1245, 1347] + list(range(1060, 1060 + int(sys.argv[1]))):
    try:
        F.test_fuzz_sketch_and_nodes(seed)
    except AssertionError as e:
        bad.append((seed, repr(e)[:60]))
print("bad", bad)
