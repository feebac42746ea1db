#!/usr/bin/env python3
"""Cycle-weighted VALU mix of an ISA line range (rates: profiles/r01_g_valu_rates.txt).
usage: mix.py file.s first last   (1-based line numbers, inclusive)"""
import re, sys
FULL = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32",
        "v_not_b32", "v_mov_b32", "v_fma_f32", "v_fmac_f32", "v_xnor_b32", "v_cndmask_b32", "v_bfi_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
def cost(op, line):
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    dpp = "_dpp" in op or "_sdwa" in op or "row_" in line or "wave_" in line
    sg = bool(re.search(r"[ ,]s\d+|[ ,]s\[|vcc|exec|0x[0-9a-f]{3,}", line.split(None, 1)[1] if " " in line.strip() or "\t" in line else ""))
    if base in FULL and not dpp:
        # an SGPR / literal operand halves the rate for the logic ops (measured for v_and sgpr, v_bitop3 sgpr); literals on v_and are full rate
        if re.search(r"[ ,]s\d+|[ ,]s\[", line): return 4.2
        return 2.4
    if base.startswith("v_cmp"): return 2.4
    return 4.2
def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(f).read().split("\n")[a - 1:b]
    n = {}; cyc = 0.0; nv = 0; ns = 0; nl = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "#")) or s.endswith(":"): continue
        op = s.split()[0]
        if op.startswith("v_"):
            c = cost(op, s); cyc += c; nv += 1
            n[op] = n.get(op, 0) + 1
        elif op.startswith("s_"): ns += 1
        elif op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")): nl += 1
    print("VALU %d  (~%.0f cycles, %.2f avg)  SALU %d  MEM/LDS %d" % (nv, cyc, cyc / max(nv, 1), ns, nl))
    for k, v in sorted(n.items(), key=lambda kv: -kv[1])[:40]: print("  %-28s %d" % (k, v))
main()
