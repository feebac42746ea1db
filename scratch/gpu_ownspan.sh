#!/bin/bash
# insert_windows_kernel: workgroup span (LDS per workgroup -> resident workgroups per CU); variants built by scratch/build_variant.sh -DOWN_SPAN_V=...
for v in base span1024 span4096; do
  [ $v != base ] && cp scratch/variants/$v.so rust_mdbg_amd/libmdbg_hip.so
  echo -n "$v: "; python scratch/measure_insert_exp.py 2>&1 | tail -1
done
