#!/bin/bash
# libmdbg_emit.so (reader, packer, emitters) under AddressSanitizer + UBSan: the unit tests and the reader fuzzers (with chunks / look-ahead margins of a few bytes).
# Swaps the in-tree library for the instrumented one and puts the product build back.  CPU only.
set -e
R=$(cd $(dirname $0)/.. && pwd); cd $R
g++ -O1 -g -std=c++17 -fPIC -Wall -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o /tmp/libmdbg_emit_asan.so rust_mdbg_amd/csrc/mdbg_emit.cpp -lz
cp rust_mdbg_amd/libmdbg_emit.so /tmp/libmdbg_emit_prod.so; trap 'cp /tmp/libmdbg_emit_prod.so rust_mdbg_amd/libmdbg_emit.so' EXIT
cp /tmp/libmdbg_emit_asan.so rust_mdbg_amd/libmdbg_emit.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_reader_cpu.py tests/test_emit_cpu.py -x -q -p no:cacheprovider | tail -2
MDBG_READER_CHUNK_BYTES=48 MDBG_READER_MARGIN_BYTES=7 python scratch/fuzz_reader.py 42 | tail -1
MDBG_READER_CHUNK_BYTES=300 MDBG_READER_MARGIN_BYTES=1 python scratch/fuzz_reader.py 43 | tail -1
python scratch/fuzz_reader.py 41 | tail -1
MDBG_READER_CHUNK_BYTES=100 MDBG_READER_MARGIN_BYTES=9 python scratch/fuzz_reader_gz.py 9 | tail -1
