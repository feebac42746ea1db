"""seeds [a, b) of test_fuzz_sketch_and_nodes in ONE process, traceback of the failures: python scratch/dbg_fuzz_range.py a b"""
import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_fuzz as F
a, b = int(sys.argv[1]), int(sys.argv[2])
for seed in range(a, b):
    try:
        F.test_fuzz_sketch_and_nodes(seed)
    except AssertionError:
        print("FAIL seed", seed); traceback.print_exc(limit=4)
print("done")
