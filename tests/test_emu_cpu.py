"""CPU checks of the sketch kernel's arithmetic: tests/emu/sketch_emu.cpp runs the very header the HIP kernel includes
(rust_mdbg_amd/csrc/bs_core.h: bit-plane compaction, bit-sliced ntHash filter, exact table evaluation, coordinate maps) with
the kernel's tile layout on the host and is compared with the oracle, minimizer by minimizer.  Test infrastructure only."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(EMU_DIR, "libsketch_emu.so")
    src = [os.path.join(EMU_DIR, "sketch_emu.cpp"), os.path.join(ROOT, "rust_mdbg_amd", "csrc", "bs_core.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src[0]])
    L = C.CDLL(so)
    L.emu_sketch.restype = C.c_int64
    L.emu_sketch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.emu_select_msb.restype = C.c_uint32
    L.emu_select_msb.argtypes = [C.c_uint32, C.c_uint32]
    L.emu_compress2.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return L


def run_emu(L, reads, l, d, hpc=True):
    b, o = O.concat_reads(reads)
    cap = len(b) + 16
    h, p, r = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    ns, nc = C.c_uint64(0), C.c_uint64(0)
    n = L.emu_sketch(b.ctypes.data, len(b), o.ctypes.data, len(o) - 1, l, d, int(hpc), h.ctypes.data, p.ctypes.data, r.ctypes.data, cap,
                     C.byref(ns), C.byref(nc))
    exp = O.sketch(b, o, l, d, already_hpc=not hpc)
    er = np.repeat(np.arange(len(o) - 1, dtype=np.uint32), np.diff(exp["off"]).astype(np.int64))
    assert n == len(exp["hashes"])
    assert np.array_equal(h[:n], exp["hashes"]) and np.array_equal(p[:n].astype(np.uint64), exp["pos"].astype(np.uint64)) and np.array_equal(r[:n], er)
    return ns.value, nc.value, n


def rnd(rng, n):
    return rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()


def test_compress_and_select_primitives(emu):
    rnd_ = random.Random(3)
    for _ in range(2000):
        m = rnd_.getrandbits(32) if rnd_.random() < 0.8 else rnd_.choice([0, 0xFFFFFFFF, 1, 0x80000000])
        x0, x1 = rnd_.getrandbits(32), rnd_.getrandbits(32)
        a, b = C.c_uint32(x0), C.c_uint32(x1)
        emu.emu_compress2(m, C.byref(a), C.byref(b))
        e0 = e1 = 0; n = 0
        for i in range(32):                       # MSB first
            if (m >> (31 - i)) & 1:
                e0 |= ((x0 >> (31 - i)) & 1) << (31 - n); e1 |= ((x1 >> (31 - i)) & 1) << (31 - n); n += 1
        assert (a.value, b.value) == (e0, e1)
        pos = [i for i in range(32) if (m >> (31 - i)) & 1]
        for j, q in enumerate(pos):
            assert emu.emu_select_msb(m, j) == q


@pytest.mark.parametrize("l", [2, 3, 5, 8, 11, 12, 14, 15, 16, 20, 23, 24, 27, 31, 32])
def test_emulated_kernel_all_l(emu, l):
    rng = np.random.default_rng(l)
    ns, nc, n = run_emu(emu, [rnd(rng, 40000), rnd(rng, 33000), b"", rnd(rng, 7)], l, 0.01)
    assert ns == 0 and (n > 500 or l < 5)


@pytest.mark.parametrize("d", [0, 1e-30, 0.0008, 0.002, 0.003, 0.03, 0.1, 0.5, 1.0, 1.5])
def test_emulated_kernel_densities(emu, d):
    rng = np.random.default_rng(7)
    ns, nc, n = run_emu(emu, [rnd(rng, 40000), rnd(rng, 3000)], 12, d)
    assert ns == 0
    if 0 < d < 0.01:
        assert nc < 4 * n + 600            # the bit-sliced filter passes few false candidates (2^-8 per strand at most)


def test_emulated_kernel_fuzz(emu):
    rnd_ = random.Random(11)
    rng = np.random.default_rng(11)

    def lowc(n):
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd_.choice(b"ACGT")]) * rnd_.choice([1, 1, 1, 2, 3, 5, 20, 300, 700])
        return bytes(out[:n])
    for it in range(30):
        reads = []
        for _ in range(rnd_.randint(1, 10)):
            n = rnd_.choice([0, 1, 5, 13, 40, 200, 3000, 20000, 33000, 70000])
            kind = rnd_.random()
            if kind < 0.5:
                reads.append(rnd(rng, n))
            elif kind < 0.8:
                reads.append(lowc(n))
            else:
                x = bytearray(rnd(rng, n))
                for _ in range(rnd_.randint(0, 3)):
                    if n:
                        x[rnd_.randrange(n)] = ord("N")
                reads.append(bytes(x))
        run_emu(emu, reads, rnd_.choice([4, 7, 10, 12, 14, 17, 21, 31, 32]), rnd_.choice([0.001, 0.003, 0.02, 0.1, 0.3]), hpc=rnd_.random() < 0.8)
    run_emu(emu, [rnd(rng, rnd_.randint(0, 150)) for _ in range(3000)], 10, 0.05)                    # many short reads per tile
    ns, _, _ = run_emu(emu, [rnd(rng, 5000) + b"A" * 40000 + rnd(rng, 5000), rnd(rng, 100)], 12, 0.01)   # homopolymer longer than a tile
    assert ns >= 1
