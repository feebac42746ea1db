#!/bin/bash
# which part of insert_windows_kernel costs what (MDBG_INSERT_EXP bit mask: 1 no key compare, 2 no ordinal cascade, 4 no count, 8 cheap hash, 16 no table access at all)
for X in 0 1 2 4 6 7 8 15 16 24; do MDBG_INSERT_EXP=$X timeout 200 python scratch/measure_insert_exp.py 2>&1 | tail -1; done
