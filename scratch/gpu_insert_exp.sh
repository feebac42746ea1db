#!/bin/bash
# (the MDBG_INSERT_EXP switches this script drives were taken out of insert_windows_kernel again after the measurement: profiles/r03_notes.md has the
# numbers, `git log -S MDBG_INSERT_EXP` the patch)
for X in 0 32 0 32; do MDBG_INSERT_EXP=$X timeout 200 python scratch/measure_insert_exp.py 2>&1 | tail -1; done
