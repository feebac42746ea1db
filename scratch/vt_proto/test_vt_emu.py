"""CPU checks of the vertical bit-slice prototype (vt_core.h; measured in round 5 and rejected: profiles/r05_b_vertical_prototype.txt).
Not part of the test suite: run with `python -m pytest scratch/vt_proto/test_vt_emu.py`."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

EMU_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(EMU_DIR))


@pytest.fixture(scope="module")
def vt():
    so = os.path.join(EMU_DIR, "libvt_emu.so")
    src = [os.path.join(EMU_DIR, "vt_emu.cpp"), os.path.join(EMU_DIR, "vt_core.h"), os.path.join(ROOT, "rust_mdbg_amd", "csrc", "bs_core.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src[0]])
    L = C.CDLL(so)
    L.vt_emu_flags.restype = C.c_long
    L.vt_emu_flags.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.vt_emu_transpose32.argtypes = [C.c_void_p]
    L.vt_emu_bits.restype = C.c_uint32
    return L


def test_vertical_transpose(vt):
    rng = np.random.default_rng(2)
    x = rng.integers(0, 1 << 32, 32, dtype=np.uint64).astype(np.uint32)
    y = x.copy()
    vt.vt_emu_transpose32(y.ctypes.data)
    for r in range(32):
        for i in range(32):
            assert (int(y[r]) >> i) & 1 == (int(x[i]) >> r) & 1


@pytest.mark.parametrize("l", [8, 12, 14, 20])
def test_vertical_filter_flags_exactly_the_lmers_with_clear_top_bits(vt, l):
    """32 machines on 32 consecutive segments of one sequence with homopolymer runs: machine i >= 1 (warmed up on segment i - 1) flags, at a kept position p of its
    segment, exactly the l-mers of the homopolymer-compressed sequence that END VT_B - 1 kept positions in front of p and whose forward or reverse ntHash has its
    top VT_B bits clear (the necessary condition for hash <= bound the sketch kernel's filter evaluates, src/read.rs:196 + nthash)"""
    B = int(vt.vt_emu_bits())
    rng = np.random.default_rng(l)
    seg_len, n_seg = 160, 32
    codes = rng.integers(0, 4, seg_len * n_seg).astype(np.uint8)
    for _ in range(120):                                   # homopolymer runs of 2 .. 12
        a = int(rng.integers(0, len(codes) - 12)); codes[a:a + int(rng.integers(2, 13))] = codes[a]
    flags = np.zeros(len(codes), np.uint8)
    assert vt.vt_emu_flags(codes.ctypes.data, seg_len, n_seg, l, flags.ctypes.data) == len(codes)
    # plain evaluation over the compressed sequence: code -> ASCII (A=0 C=1 T=2 G=3), oracle's canonical hashes are min(fwd, rev): recompute both strands here
    seed = {0: 0x3c8bfbb395c60474, 1: 0x3193c18562a02b4c, 3: 0x20323ed082572324, 2: 0x295549f54be24456}
    comp = {0: 2, 1: 3, 2: 0, 3: 1}
    rol = lambda x, r: ((x << (r % 64)) | (x >> ((64 - r) % 64))) & ((1 << 64) - 1) if r % 64 else x
    kept = [p for p in range(len(codes)) if p == 0 or codes[p] != codes[p - 1]]
    dense = [int(codes[p]) for p in kept]
    index_of = {p: j for j, p in enumerate(kept)}
    want = np.zeros(len(codes), np.uint8)
    for j in range(len(dense)):
        e = j - (B - 1)                                   # the l-mer ends B - 1 kept positions in front of kept position j
        if e - (l - 1) < 0:
            continue
        fh = rh = 0
        for u in range(l):
            c = dense[e - u]
            fh ^= rol(seed[c], u); rh ^= rol(seed[comp[c]], l - 1 - u)
        if (fh >> (64 - B)) == 0 or (rh >> (64 - B)) == 0:
            want[kept[j]] = 1
    # machine i owns segment i; machine 0 has no warm-up data: its segment is not compared.  Every flag sits on a kept position.
    own = slice(seg_len, len(codes))
    assert not np.any(flags[[p for p in range(len(codes)) if p not in index_of]])
    assert np.array_equal(flags[own], want[own]) and int(want[own].sum()) >= 3
