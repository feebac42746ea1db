#!/bin/bash
set -u
for i in 1 2 3 4 5; do timeout 600 python scratch/anchor_alone.py 2>/dev/null | tail -1; done
