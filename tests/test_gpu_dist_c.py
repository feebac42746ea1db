"""The multi-GPU layer behind the C ABI (include/mdbg_dist.h): a plain C program drives several ranks (threads sharing the GPU, its own
communicator), and the RCCL transport is exercised with a real one-rank ncclComm_t."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "rust_mdbg_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("distc") / "mdbg_dist_threads")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mdbg_dist_threads.c"),
                    "-L" + LIB, "-lmdbg_hip", "-lpthread", "-Wl,-rpath," + LIB, "-o", out], check=True)
    return out


@pytest.mark.parametrize("world,reads,rounds,packed,chunks", [(1, 300, 1, 0, 1), (2, 300, 2, 0, 1), (3, 300, 3, 1, 1), (4, 200, 1, 0, 1), (8, 160, 2, 1, 1),
                                                              (1, 300, 1, 0, 3), (2, 300, 2, 0, 4), (3, 300, 1, 1, 2), (8, 160, 2, 1, 5), (2, 40, 1, 0, 64)])
def test_c_program_drives_ranks_through_mdbg_dist(exe, world, reads, rounds, packed, chunks):
    """chunks > 1: mdbg_dist_set_pipeline cuts every ingest call into that many rounds (more chunks than reads: empty rounds)"""
    r = subprocess.run([exe, str(world), str(reads), str(rounds), str(packed), str(chunks)], capture_output=True, text=True, timeout=90)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "EQUAL to the single-context table" in r.stdout


def _bytes_in(stdout):
    return int(stdout.split("bytes received by all ranks: ")[1].split()[0].rstrip(";"))


@pytest.mark.parametrize("world,reads,rounds,packed,chunks", [(2, 300, 2, 0, 1), (8, 160, 2, 1, 1), (3, 300, 1, 1, 2)])
def test_whole_sketch_exchange_gives_the_same_table_and_moves_more(exe, world, reads, rounds, packed, chunks):
    """mdbg_dist_set_exchange: the default ships per peer the window list and only the hashes its windows need (segments); WHOLE ships every
    sketch to every rank.  Same table either way; from four ranks on the segments are the smaller exchange by a wide margin (k = 9 here: a run
    of r windows needs r + 8 hashes)"""
    out = {}
    for whole in (0, 1):
        r = subprocess.run([exe, str(world), str(reads), str(rounds), str(packed), str(chunks), str(whole)], capture_output=True, text=True, timeout=90)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "EQUAL to the single-context table" in r.stdout and ("whole-sketch exchange" if whole else "segment exchange") in r.stdout
        out[whole] = _bytes_in(r.stdout)
        # re-windowing the resident sketches at another k: the resident hashes do with whole sketches; under segments the rounds are exchanged again for the new k
        # (round 5; until then MDBG_E_STATE) — tests/test_gpu_round5.py compares the tables
        assert "reset(k + 2): 0 " in r.stdout and "ranks differ" not in r.stdout, r.stdout
    assert out[0] > 0 and out[1] > 0
    if world >= 8:
        assert out[0] < 0.6 * out[1], out


@pytest.fixture(scope="module")
def exe_procs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("distp") / "mdbg_dist_procs")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mdbg_dist_procs.c"),
                    "-L" + LIB, "-lmdbg_hip", "-lpthread", "-Wl,-rpath," + LIB, "-o", out], check=True)
    return out


@pytest.mark.parametrize("world,reads,rounds,packed,chunks", [(2, 300, 2, 0, 1), (2, 300, 1, 1, 3), (3, 300, 3, 0, 2), (4, 200, 2, 1, 1)])
def test_separate_processes_drive_ranks_through_mdbg_dist(exe_procs, world, reads, rounds, packed, chunks):
    """the ranks are OS processes (forked before the GPU runtime is touched), each with its own library state and mdbg_dist, sharing the
    one GPU; the communicator is a shared-memory function table (host-staged exchange).  Partitions put together == single-context table"""
    r = subprocess.run([exe_procs, str(world), str(reads), str(rounds), str(packed), str(chunks)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "world %d PROCESSES" % world in r.stdout and "EQUAL to the single-context table" in r.stdout


from rust_mdbg_amd.dist_c import Comm      # noqa: E402  (mirror of mdbg_comm)


@pytest.mark.parametrize("chunks", [1, 3])
def test_rccl_transport_with_a_one_rank_communicator(chunks):
    """mdbg_comm_rccl on a real ncclComm_t (1 rank: all-gather, an empty send/recv group and the all-reduce run through RCCL); the table
    equals the plain single-GPU one"""
    import torch  # noqa: F401  (brings torch's RCCL into the process: the library resolves the nccl* symbols from it)
    import rust_mdbg_amd as R
    from rust_mdbg_amd import api
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    class UniqueId(C.Structure):            # ncclUniqueId is passed BY VALUE (rccl.h:220)
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    L = api.load_library()
    vt = Comm()
    L.mdbg_comm_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Comm)]
    assert L.mdbg_comm_rccl(comm, 0, 1, C.byref(vt)) == 0
    k, l, d, a, n_reads = 15, 12, 0.004, 2, 3000
    P = api.Params(k=k, l=l, density=d, min_abundance=a, reads_already_hpc=0, device=-1, flags=0, table_capacity_hint=0)
    err = C.c_int()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(Comm), C.POINTER(C.c_int)]
    dd = L.mdbg_dist_create(C.byref(P), C.byref(vt), C.byref(err))
    assert dd and err.value == 0
    assert vt.exchange_begin and vt.exchange_wait                      # the RCCL transport has the split form
    L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
    assert L.mdbg_dist_set_pipeline(dd, chunks) == 0 and L.mdbg_dist_set_pipeline(dd, 0) == -1 and L.mdbg_dist_set_pipeline(dd, 65) == -1
    L.mdbg_dist_ingest_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    with R.Mdbg(k, l, d, a) as m:
        db, do, nb = m.synth_reads_device(seed=3, genome_len=4_000_000, n_reads=n_reads)
        m.ingest_device(db, do, n_reads, nb, 0)
        ref = m.finalize()
        half = n_reads // 2
        offs = m.to_host(do, (n_reads + 1) * 8, np.uint64)
        cut = int(offs[half]) // 16 * 16
        o2 = torch.from_numpy((offs[half:] - cut).astype(np.int64)).cuda()
        torch.cuda.synchronize()
        assert L.mdbg_dist_ingest_batch_device(dd, db, do, half, int(offs[half]), 0) == 0           # two rounds
        assert L.mdbg_dist_ingest_batch_device(dd, db + cut, o2.data_ptr(), n_reads - half, nb - cut, half) == 0
        nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
        assert L.mdbg_dist_finalize(dd, C.byref(nd), C.byref(row), C.byref(ng)) == 0
        n = int(nd.n)
        assert n == ng.value == ref["n_nodes"] > 1000 and int(nd.n_distinct) == ref["n_nodes_before"]
        keys = m.to_host(C.cast(nd.keys, C.c_void_p).value, n * k * 8, np.uint64).reshape(n, k)
        rows = m.to_host(row.value, n * 8, np.uint64)
        index = m.to_host(C.cast(nd.index, C.c_void_p).value, n * 4, np.uint32)
        assert np.array_equal(keys[np.argsort(rows)], ref["keys"]) and np.array_equal(index[np.argsort(rows)], ref["index"])
    L.mdbg_dist_destroy(dd)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_pipelined_packed_batch_with_exceptions_through_the_c_layer():
    """mdbg_dist_set_pipeline on a 2-bit packed batch that carries an exception list (N inside reads): every chunk gets its slice of the
    list, rebased to the chunk's first base; the table equals the single-context one"""
    import torch
    import rust_mdbg_amd as R
    from rust_mdbg_amd import api, emit as E, synth
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    from rust_mdbg_amd.dist_c import UniqueId
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    L = api.load_library()
    vt = Comm()
    L.mdbg_comm_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Comm)]
    assert L.mdbg_comm_rccl(comm, 0, 1, C.byref(vt)) == 0
    k, l, d, a = 9, 12, 0.004, 2
    reads = synth.synth_reads(11, 300000, 400, mean_len=7000, sd_len=2000, min_len=300, max_len=14000, err_ppm=2000)
    reads = [r[:len(r) // 3] + b"N" + r[len(r) // 3 + 1:] if i % 5 == 0 and len(r) > 40 else r for i, r in enumerate(reads)]
    from oracle import oracle as O
    bases, offs = O.concat_reads(reads)
    pk = E.pack_reads(bases, offs)
    assert len(pk["exc_pos"]) == 80
    with R.Mdbg(k, l, d, a) as m:
        m.ingest_packed(pk, 0)
        ref = m.finalize()
    P = api.Params(k=k, l=l, density=d, min_abundance=a, reads_already_hpc=0, device=-1, flags=0, table_capacity_hint=0)
    err = C.c_int()
    L.mdbg_dist_create.restype = C.c_void_p
    L.mdbg_dist_create.argtypes = [C.POINTER(api.Params), C.POINTER(Comm), C.POINTER(C.c_int)]
    L.mdbg_dist_set_pipeline.argtypes = [C.c_void_p, C.c_uint32]
    L.mdbg_dist_ingest_batch_packed_device.argtypes = [C.c_void_p, C.POINTER(api.PackedBatch), C.c_uint64, C.c_uint64]
    L.mdbg_dist_finalize.argtypes = [C.c_void_p, C.POINTER(api.Nodes), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mdbg_dist_destroy.argtypes = [C.c_void_p]
    L.mdbg_dist_ctx.restype = C.c_void_p
    L.mdbg_dist_ctx.argtypes = [C.c_void_p]
    for chunks in (3, 7):
        if chunks != 3:
            assert L.mdbg_comm_rccl(comm, 0, 1, C.byref(vt)) == 0          # one transport object per mdbg_dist (mdbg_dist_destroy frees it)
        dd = L.mdbg_dist_create(C.byref(P), C.byref(vt), C.byref(err))
        assert dd and err.value == 0 and L.mdbg_dist_set_pipeline(dd, chunks) == 0
        tw = torch.from_numpy(pk["words"].view(np.int64)).cuda()
        to = torch.from_numpy(pk["offsets"].view(np.int64)).cuda()
        tp = torch.from_numpy(pk["exc_pos"].view(np.int64)).cuda()
        tv = torch.from_numpy(pk["exc_val"]).cuda()
        torch.cuda.synchronize()
        pb = api.PackedBatch(tw.data_ptr(), to.data_ptr(), len(reads), tp.data_ptr(), tv.data_ptr(), len(pk["exc_pos"]))
        assert L.mdbg_dist_ingest_batch_packed_device(dd, C.byref(pb), len(bases), 0) == 0
        nd, row, ng = api.Nodes(), C.c_void_p(), C.c_uint64()
        assert L.mdbg_dist_finalize(dd, C.byref(nd), C.byref(row), C.byref(ng)) == 0
        n = int(nd.n)
        assert n == ng.value == ref["n_nodes"] > 500 and int(nd.n_distinct) == ref["n_nodes_before"]
        with R.Mdbg(k, l, d, a) as m:           # any context can copy device memory to the host
            keys = m.to_host(C.cast(nd.keys, C.c_void_p).value, n * k * 8, np.uint64).reshape(n, k)
            rows = m.to_host(row.value, n * 8, np.uint64)
            abund = m.to_host(C.cast(nd.abundance, C.c_void_p).value, n * 2, np.uint16)
        assert np.array_equal(keys[np.argsort(rows)], ref["keys"]) and np.array_equal(abund[np.argsort(rows)], ref["abundance"])
        L.mdbg_dist_destroy(dd)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
