#!/usr/bin/env python3
"""Instruction-ISSUE roofline of the tile kernel (sketch_bs_kernel<l>): how many VALU-pipe cycles does one launch need, and how many does it get?

The SQ counters give instruction COUNTS per type (SQ_INSTS_VALU, ...); none of them splits the VALU count by issue cost (tried this round:
SQ_THREAD_CYCLES_VALU = SQ_INSTS_VALU x 64 for a v_xor stream and a v_alignbit stream alike, profiles/r05_a_thread_cycles_probe.txt; PC sampling is
not supported on the pool's boxes, ATT has no decoder library in the image).  So the split comes from the ISA:

  1. the kernel is compiled once more from a patched COPY of csrc/ (build_probe: MDBG_HOT(cond, value) yields `value`, i.e. the run-time conditions that are constant
     on the benchmark's workload become compile-time constants, the phase stamps become marker lines), so the ISA of the probe build IS the hot path;
  2. every instruction is classed (full-rate VALU / half-rate VALU / SALU / LDS / VMEM / SMEM / control) with the rules measured in
     profiles/r01_g_valu_rates.txt (half rate: v_alignbit, v_perm, shifts left, v_bfe, v_bcnt, v_mbcnt, v_min/max, v_cmp, v_cndmask, multiplies,
     three-operand integer ops other than v_bitop3, packed 16-bit, DPP / SDWA forms, v_bfrev, 64-bit shifts, ANY VALU instruction with an SGPR source);
  3. per phase (marker to marker) the static counts are weighted with the trip counts that are known (the filter loop: 3.25 steps per wave on this
     workload) and then SCALED to the phase's dynamic VALU count from the counters (MDBG_STOP_PHASE passes, scratch/gpu_r5_a.sh) — the scale factor is
     printed: 1.0 means the static path reproduces the counter;
  4. issue cycles = sum over phases of dynamic VALU count x (share_full x c_full + share_half x c_half), with c_* the issue cycles per wave64
     instruction and SIMD measured at the kernel's occupancy (profiles/r05_a_valu_rate_occupancy.txt, 6 waves per SIMD).

usage: profiles/isa_issue.py [--l 12] [--phase-counters gpurun_out/r5a/phase_counters.txt] [--out profiles/r05_isa_issue.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FULL = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_bitop3_b32", "v_fma_f32", "v_fmac_f32", "v_xnor_b32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_i32", "v_rcp_iflag_f32",
        "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def classify(mn, ops, line):
    """-> one of full, half, salu, smem, lds, vmem, ctrl"""
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("s_"):
        if mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
            return "smem"
        if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_sleep", "s_setprio", "s_sethalt", "s_set_gpr_idx")):
            return "ctrl"
        return "salu"
    if not mn.startswith("v_"):
        return "ctrl"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if mn.endswith(("_dpp", "_sdwa")) or "row_" in line or "quad_perm" in line or "wave_sh" in line or "row_bcast" in line:
        return "half"
    if base not in FULL:
        return "half"
    # an SGPR (or vcc / exec / m0) SOURCE makes it half rate; operand 0 is the destination
    for o in ops[1:]:
        o = o.strip()
        if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)$", o):
            return "half"
    return "full"


def parse_kernel(path, l):
    txt = open(path).read()
    head = "_Z16sketch_bs_kernelILi%dELi0ELi1ELi4ELi1EEv10SketchArgs:" % l
    i = txt.index(head)
    body = txt[i:]
    body = body[:body.index(".Lfunc_end")]
    insts = []          # (line_no, mnemonic, klass, block, loop_header, depth)
    phase = -1
    block, loop, depth = "entry", None, 0
    for n, ln in enumerate(body.split("\n")):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", ln)
        if m:
            block = m.group(1)
            c = m.group(2)
            mh = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
            if "Loop Header" in c:
                md = re.search(r"Depth=(\d+)", c)
                loop, depth = block[2:], int(md.group(1))
            elif mh:
                loop, depth = mh.group(1), int(mh.group(2))
            else:
                loop, depth = None, 0
            continue
        m = re.match(r"^\t([a-z][a-z0-9_]+)\s*(.*)$", ln)
        if not m:
            continue
        mn, rest = m.group(1), m.group(2)
        if "MDBG_PHASE_MARK" in rest:
            phase = int(rest.split("MDBG_PHASE_MARK")[1].split()[0])
            continue
        ops = [o for o in rest.split(";")[0].split(",")] if rest else []
        insts.append((n, mn, classify(mn, ops, rest), block, loop, depth, phase))
    return insts


def build_probe(l, asm):
    """the probe: a COPY of csrc/ in a temporary directory in which MDBG_HOT(cond, value) yields `value` (the run-time conditions that are constant on the
    benchmark's workload become compile-time constants) and the phase stamps become marker lines — the product sources have no switch for this"""
    import shutil
    import tempfile
    src = os.path.join(ROOT, "rust_mdbg_amd", "csrc")
    tmp = tempfile.mkdtemp(prefix="mdbg_isa_probe_")
    work = os.path.join(tmp, "rust_mdbg_amd", "csrc")
    shutil.copytree(src, work, ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(work, "sketch.hip")
    t = open(p).read()
    hot, stamp = "#define MDBG_HOT(cond, value) (cond)", "#define MDBG_STAMP(i) do {"
    assert t.count(hot) == 1 and t.count(stamp) == 1, "sketch.hip: the MDBG_HOT / MDBG_STAMP definitions moved"
    t = t.replace(hot, "#define MDBG_HOT(cond, value) (value)")
    a = t.index(stamp); b = t.index("\n", a)
    t = t[:a] + '#define MDBG_STAMP(i) asm volatile("s_nop 0 ; MDBG_PHASE_MARK " #i ::: "memory")' + t[b:]
    open(p, "w").write(t)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-DMDBG_ONLY_L=%d" % l,
                           "-o", asm, "libmdbg.hip"], cwd=work, stderr=subprocess.DEVNULL)
    shutil.rmtree(tmp, ignore_errors=True)
    return asm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--l", type=int, default=12)
    ap.add_argument("--phase-counters", default=os.path.join(ROOT, "profiles", "r05_a_phase_counters.txt"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_isa_issue.json"))
    ap.add_argument("--asm", default=None, help="an existing probe .s (default: build it)")
    ap.add_argument("--c-full", type=float, default=2.63, help="issue cycles of a full-rate VALU instruction at 6 waves per SIMD (v_xor 2.59, v_bitop3 2.67)")
    ap.add_argument("--c-half", type=float, default=4.5, help="... of a half-rate one (v_alignbit 4.43, v_perm 4.64)")
    ap.add_argument("--bases-per-launch", type=float, default=6999190596.0, help="raw bases one launch of the counter passes covered (bench.py default workload)")
    ap.add_argument("--density", type=float, default=0.002)
    ap.add_argument("--filter-trips", type=float, default=3.25, help="filter steps per wave (13 steps of 63 dense words over 4 waves on configs[2])")
    args = ap.parse_args()
    asm = args.asm
    if asm is None:
        asm = "/tmp/mdbg_isa_probe_l%d.s" % args.l
        asm = build_probe(args.l, asm)
    insts = parse_kernel(asm, args.l)
    # dynamic counts per phase from the early-exit passes: cumulative -> per phase; per launch
    cum = {}
    for ln in open(args.phase_counters):
        m = re.match(r"stop_after (\d+) launches (\d+) ms (\[.*?\]) (\{.*\})", ln)
        if m:
            d = {k: float(v) for k, v in eval(m.group(4)).items()}
            cum[int(m.group(1))] = {k: v / int(m.group(2)) for k, v in d.items()}
    order = [1, 2, 3, 0]                    # stop after phase 1, 2, 3, complete kernel
    names = {1: "1 load, setup", 2: "2 keep masks, compaction, dense stream", 3: "3 bit-sliced filter", 0: "4 exact, ranks, records"}
    prev = collections.defaultdict(float)
    dyn = {}
    for p in order:
        dyn[p] = {k: cum[p][k] - prev[k] for k in cum[p] if k.startswith("SQ_INSTS")}
        prev = cum[p]
    # static weighted histogram per phase.  ISA phase marks: 0..1 = phase 1, 1..2 = phase 2, 2..3 = phase 3, >= 3 = phase 4
    mark_to_phase = lambda mk: 1 if mk <= 0 else 2 if mk == 1 else 3 if mk == 2 else 0
    # the filter loop: the loop of phase 3 with the most VALU instructions
    loop_valu = collections.Counter()
    for (_, mn, k, blk, loop, depth, mk) in insts:
        if mark_to_phase(mk) == 3 and loop and k in ("full", "half"):
            loop_valu[loop] += 1
    filter_loop = loop_valu.most_common(1)[0][0] if loop_valu else None
    stat = {p: collections.Counter() for p in order}
    stat_loop = collections.Counter()
    for (_, mn, k, blk, loop, depth, mk) in insts:
        p = mark_to_phase(mk)
        w = args.filter_trips if (p == 3 and loop == filter_loop) else 1.0
        stat[p][k] += w
        if p == 3 and loop == filter_loop:
            stat_loop[k] += 1
    waves = cum[0].get("waves", 861124.0)
    out = {"note": __doc__.split("\n")[0], "l": args.l, "config": {"l": args.l, "density": args.density, "input_format": "packed", "bases_per_launch": args.bases_per_launch}, "c_full": args.c_full, "c_half": args.c_half, "filter_trips_per_wave": args.filter_trips,
           "filter_loop_body": dict(stat_loop), "phases": {}, "asm_lines": len(insts)}
    tot_cycles = tot_valu = tot_full = tot_half = 0.0
    for p in order:
        s = stat[p]
        sv = s["full"] + s["half"]
        dv = dyn[p]["SQ_INSTS_VALU"]
        share_half = s["half"] / sv if sv else 0.0
        cyc = dv * ((1 - share_half) * args.c_full + share_half * args.c_half)
        tot_cycles += cyc; tot_valu += dv; tot_full += dv * (1 - share_half); tot_half += dv * share_half
        out["phases"][names[p]] = {
            "static_per_wave": {k: round(v, 1) for k, v in s.items()},
            "dynamic_per_launch": {k: v for k, v in dyn[p].items()},
            "dynamic_valu_per_wave": dv / 861124.0,
            "static_over_dynamic_valu": sv / (dv / 861124.0) if dv else None,
            "half_rate_share": share_half,
            "valu_issue_cycles_per_launch": cyc}
    out["valu_per_launch"] = tot_valu
    out["valu_full_rate"] = tot_full
    out["valu_half_rate"] = tot_half
    out["half_rate_share"] = tot_half / tot_valu
    out["valu_issue_cycles_per_launch"] = tot_cycles
    out["avg_issue_cycles_per_valu"] = tot_cycles / tot_valu
    out["valu_issue_cycles_per_base"] = tot_cycles / args.bases_per_launch
    out["valu_instructions_per_base"] = tot_valu / args.bases_per_launch
    other = {k: cum[0][k] for k in cum[0] if k.startswith("SQ_INSTS") and k != "SQ_INSTS_VALU"}
    out["other_instructions_per_launch"] = other
    out["all_instructions_per_launch"] = tot_valu + sum(other.values())
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "phases"}, indent=1))
    for n, ph in out["phases"].items():
        print("%-42s static/dynamic VALU %.2f  half-rate share %.3f  dynamic VALU per wave %.1f" % (n, ph["static_over_dynamic_valu"], ph["half_rate_share"], ph["dynamic_valu_per_wave"]))


if __name__ == "__main__":
    sys.exit(main())
