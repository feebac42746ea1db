#!/bin/bash
for X in 0 32 0 32; do MDBG_INSERT_EXP=$X timeout 200 python scratch/measure_insert_exp.py 2>&1 | tail -1; done
