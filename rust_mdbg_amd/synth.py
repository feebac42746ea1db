"""CPU (numpy) regenerator of the device-side synthetic read generator (csrc/synth.hip): identical bytes.

Integer-only and counter-based: byte = f(seed, read index, position).  Used to build test inputs that the oracle
can consume and to cross-check mdbg_synth_reads_device.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def sm64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def rnd3(seed, a, b):
    """sm64(sm64(seed ^ a) + b), all modulo 2^64; a and b may be scalars or uint64 arrays"""
    with np.errstate(over="ignore"):
        return sm64(sm64(np.uint64(seed) ^ np.asarray(a, dtype=np.uint64)) + np.asarray(b, dtype=np.uint64))


def read_geom(seed, r, genome_len, mean_len, sd_len, min_len, max_len):
    v = int(rnd3(seed + 1, np.uint64(r), 1))
    s = (v & 0xFFFF) + ((v >> 16) & 0xFFFF) + ((v >> 32) & 0xFFFF) + ((v >> 48) & 0xFFFF)
    ln = mean_len + (s * sd_len) // 37837 - (131070 * sd_len) // 37837
    ln = max(ln, min_len)
    ln = min(ln, max_len)
    ln = min(ln, genome_len)
    start = int(rnd3(seed + 1, np.uint64(r), 2)) % (genome_len - ln + 1)
    strand = int(rnd3(seed + 1, np.uint64(r), 3)) & 1
    return start, ln, strand


def synth_read(seed, r, genome_len, mean_len=15000, sd_len=1500, min_len=8000, max_len=25000, err_ppm=1000):
    start, span, strand = read_geom(seed, r, genome_len, mean_len, sd_len, min_len, max_len)
    thr24 = (err_ppm << 24) // 1000000
    i = np.arange(span, dtype=np.uint64)
    g = (rnd3(seed, np.uint64(start) + i, 0x47) >> np.uint64(62)).astype(np.int64)
    e = rnd3(seed + 2, np.full(span, r, dtype=np.uint64), i)
    err = (e >> np.uint64(40)).astype(np.int64) < thr24
    ty = ((e >> np.uint64(8)) % np.uint64(3)).astype(np.int64)
    sub = (g + 1 + ((e >> np.uint64(4)) % np.uint64(3)).astype(np.int64)) & 3
    ins = (e & np.uint64(3)).astype(np.int64)
    cnt = np.ones(span, dtype=np.int64)
    cnt[err & (ty == 1)] = 2
    cnt[err & (ty == 2)] = 0
    c0 = np.where(err & (ty == 0), sub, g)
    off = np.concatenate([[0], np.cumsum(cnt)])
    out = np.zeros(off[-1], dtype=np.int64)
    has = cnt >= 1
    out[off[:-1][has]] = c0[has]
    two = cnt == 2
    out[off[:-1][two] + 1] = ins[two]
    if strand:
        out = 3 - out[::-1]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[out].tobytes()


def synth_reads(seed, genome_len, n_reads, first_read=0, **kw):
    return [synth_read(seed, first_read + r, genome_len, **kw) for r in range(n_reads)]
