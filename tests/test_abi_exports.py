"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol include/mdbg_hip.h declares
(no compute calls — there is no GPU here)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "mdbg_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(mdbg_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    from rust_mdbg_amd import api
    L = api.load_library()
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(api.EXPORTS) == syms
    assert L.mdbg_abi_version() == 3
    assert L.mdbg_strerror(0) == b"ok" and b"ACGTN" in L.mdbg_strerror(-2)


def test_build_hook_checks_agree_with_the_tree():
    """the checks __graft_entry__.build() runs behind the compile step (it once kept an ABI version of its own)"""
    import __graft_entry__ as G
    G.check_built()


def test_emit_library_exports_every_declared_symbol():
    from rust_mdbg_amd import emit
    h = open(os.path.join(ROOT, "include", "mdbg_emit.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mdbg_(?:emit|seqfile|reader|pack|packed|lmer)_[a-z_0-9]+)\s*\(", h)))
    L = emit.load_library()
    assert syms == sorted(emit.EXPORTS + emit.READER_EXPORTS)
    for s in syms:
        assert hasattr(L, s), s


def test_struct_layouts_match_header():
    import ctypes as C
    from rust_mdbg_amd import api
    assert C.sizeof(api.Params) == 4 + 4 + 8 + 4 + 4 + 4 + 4 + 8 + 4 + 4 + 24
    assert C.sizeof(api.Stats) == 8 * 8 + 4 * 8 + 2 * 8 + 5 * 8
    assert C.sizeof(api.SynthParams) == 3 * 8 + 6 * 4
    assert api.Nodes.keys.offset == 16 and api.Nodes.n_distinct.offset == 16 + 10 * 8


def test_ctypes_mirrors_match_the_c_compiler(tmp_path):
    """sizes and field offsets of every struct the Python mirrors declare, as gcc lays out the headers' own definitions"""
    import ctypes as C
    import subprocess
    from rust_mdbg_amd import api, dist_c, emit
    src = tmp_path / "layout.c"
    fields = {
        "mdbg_params": (api.Params, ["k", "l", "density", "min_abundance", "reads_already_hpc", "device", "flags", "table_capacity_hint", "scheme", "syncmer_s"]),
        "mdbg_packed_batch": (api.PackedBatch, ["words", "offsets", "n_reads", "exc_pos", "exc_val", "n_exc"]),
        "mdbg_nodes": (api.Nodes, ["n", "k", "keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed", "n_distinct", "n_wrapped"]),
        "mdbg_stats": (api.Stats, ["n_reads", "n_minimizers", "ms_sketch", "ms_sketch_tile", "tile_bases"]),
        "mdbg_comm": (dist_c.Comm, ["self", "rank", "world", "allgather_u64", "exchange", "allreduce_sum_u64", "exchange_begin", "exchange_wait"]),
        "mdbg_edges": (emit.Edges, ["n", "n1", "o1", "n2", "o2", "overlap", "presimp_removed"]),
    }
    body = "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (t, t) + "".join('printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (t, f, t, f) for f in fs)
                   for t, (_, fs) in fields.items())
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "mdbg_dist.h"\n#include "mdbg_emit.h"\nint main(void) {\n' + body + 'return 0; }\n')
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    got = dict(line.split() for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().split("\n"))
    for t, (cls, fs) in fields.items():
        assert int(got[t]) == C.sizeof(cls), t
        for f in fs:
            assert int(got["%s.%s" % (t, f)]) == getattr(cls, f).offset, (t, f)


def test_create_without_gpu_fails_cleanly():
    """no silent CPU fallback: without a device mdbg_create reports MDBG_E_DEVICE"""
    import torch
    import pytest
    import rust_mdbg_amd as R
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(R.MdbgError) as e:
        R.Mdbg(5, 12, 0.01)
    assert e.value.code == -4
    with pytest.raises(R.MdbgError) as e:
        R.Mdbg(1, 12, 0.01)
    assert e.value.code == -1


def test_the_built_library_is_a_product_build():
    """mdbg_build_flags() == 0: the wave-tile experiment's kernels did not go into the in-tree library (the product sources carry no other experiment switch since round 6);
    the host-memory calls work without a device (ordinary memory until an ingest call sees it)"""
    from rust_mdbg_amd import api
    L = api.load_library()
    assert L.mdbg_build_flags() == 0 and L.mdbg_abi_version() > 0
    p = L.mdbg_host_alloc(1 << 16)
    assert p and p % 4096 == 0 and L.mdbg_host_is_pinned(p) == 0
    L.mdbg_host_free(p)
    L.mdbg_host_free(None)


def test_headers_are_plain_c_and_the_example_links(tmp_path):
    """include/*.h must be usable from C (the drop-in boundary is a C ABI): the plain-C example host compiles with gcc -std=c99
    and links against the two libraries (running it needs a GPU: tests/test_gpu_pipeline.py)"""
    import subprocess
    lib = os.path.join(ROOT, "rust_mdbg_amd")
    exe = str(tmp_path / "mdbg_cli")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "mdbg_cli.c"), "-L" + lib, "-lmdbg_hip", "-lmdbg_emit", "-lpthread", "-Wl,-rpath," + lib, "-o", exe], check=True)
    assert os.path.exists(exe)


def test_dist_header_symbols_and_c_example_link(tmp_path):
    """include/mdbg_dist.h: every declared entry point is exported, and the plain-C multi-rank example compiles as C99 and links
    (running it needs a GPU: tests/test_gpu_dist_c.py)"""
    import subprocess
    from rust_mdbg_amd import api
    h = open(os.path.join(ROOT, "include", "mdbg_dist.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mdbg_(?:dist|comm)_[a-z_0-9]+)\s*\(", h)))
    assert syms == ["mdbg_comm_rccl", "mdbg_dist_create", "mdbg_dist_ctx", "mdbg_dist_destroy", "mdbg_dist_finalize", "mdbg_dist_ingest_batch_device",
                    "mdbg_dist_ingest_batch_packed_device", "mdbg_dist_reset", "mdbg_dist_set_exchange", "mdbg_dist_set_pipeline", "mdbg_dist_stage_ms", "mdbg_dist_stage_name",
                    "mdbg_dist_traffic"]
    L = api.load_library()
    for s in syms:
        assert hasattr(L, s), s
    lib = os.path.join(ROOT, "rust_mdbg_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "mdbg_dist_threads.c"), "-L" + lib, "-lmdbg_hip", "-lpthread", "-Wl,-rpath," + lib,
                    "-o", str(tmp_path / "mdbg_dist_threads")], check=True)
