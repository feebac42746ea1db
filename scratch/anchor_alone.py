"""scale_anchor_n1 of bench.py in a fresh process (no default-workload legs in front of it): is the anchor's rate a property of the process state it runs in?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rust_mdbg_amd as R
import bench
r = bench.scale_anchor_n1(R, torch, np, 0, 2, oracle_shard=False)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "steps")}))
