"""bench.py's launcher contract and the RCCL id hand-over, as far as they can be checked without a GPU."""
import ctypes as C
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_gpus_n_without_n_gpus_fails_loudly():
    """`python bench.py --gpus 2` must never print an N=1 line: with fewer than 2 visible GPUs it exits non-zero and says why"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 needs 2 GPUs" in r.stderr and r.stdout.strip() == ""


def test_bench_world_size_must_match_gpus_flag():
    """a launcher that starts 2 ranks while the command line says --gpus 1 (or the reverse) is a configuration error, not a mislabelled line"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 1 but the launcher started 2 rank(s)" in r.stderr and r.stdout.strip() == ""


def test_bench_human_workload_needs_a_divisor_of_its_shards():
    """--workload human is the same eight shards at every N: a rank count that does not divide them is refused before anything runs"""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus must divide 8" in r.stderr and r.stdout.strip() == ""


def test_nccl_unique_id_travels_whole():
    """the 128-byte ncclUniqueId is binary: a NUL byte inside must not shorten what rank 0 broadcasts (rust_mdbg_amd/dist_c.py)"""
    from rust_mdbg_amd.dist_c import UniqueId
    uid = UniqueId()
    raw = bytes([7, 9, 0, 3] + [0] * 60 + list(range(1, 65)))
    C.memmove(C.byref(uid), raw, 128)
    assert bytes(uid.internal) != raw                      # the trap: c_char arrays convert like C strings
    blob = C.string_at(C.byref(uid), C.sizeof(uid))         # what rccl_comm() sends
    assert blob == raw and len(blob) == 128
    back = UniqueId()
    C.memmove(C.byref(back), blob, C.sizeof(back))
    assert C.string_at(C.byref(back), 128) == raw
