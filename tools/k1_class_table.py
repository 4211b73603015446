"""Cycle-weighted phase table of K1 from the ISA of the built library: the kernel's instructions
between its phase stamps (s_memrealtime = stamp(k) in scan_segments.h), split into the two issue classes that
tools/valu_rate.hip measures on MI355X -- FAST (v_mov, 32-bit add / sub / and / or / xor / not, right shifts:
2.4-2.7 cycles per wave64 instruction at four waves per SIMD) and SLOW (everything else: VOP3 forms, left shifts,
multiplies, dot products, permutes, packed math, compares, selects, ffbh / ffbl, SDWA / DPP: 4.3 cycles) --, and,
given the executed VALU instructions per wave and phase (PMC, tools/pmc_phases.sh), the cycles each phase occupies a
SIMD for.  The STATIC mix of a phase (all its code paths) stands in for the executed mix.
  python tools/k1_class_table.py /tmp/isa/k1.s [P1 P2 P3 P4 executed VALU per wave]"""
import re, sys
FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
C_FAST, C_SLOW = 2.7, 4.3
lines = open(sys.argv[1]).read().splitlines()
phases = [[]]
for l in lines:
    t = l.strip().split()
    if not t: continue
    op = t[0]
    if op == "s_memrealtime":                       # stamp(k): a phase boundary (the cycle-counter twin s_memtime sits beside it)
        phases.append([])
        continue
    phases[-1].append((op, l))
names = ["prologue", "P1 colour", "P2 fDCT + quantize", "P3 DC + sort", "P3 walks", "P4 scan", "P4 stitch", "P4 flush", "end"]
def cls(op, l):
    if not op.startswith("v_"): return None
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith(("_e64", "_sdwa", "_dpp")) or "dpp" in l or "sdwa" in l.lower(): return "slow"
    return "fast" if base in FAST else "slow"
tot = []
print("%-20s %6s %6s %6s %6s %6s  fast share" % ("phase (static)", "VALU", "fast", "slow", "SALU", "LDS"))
for i, ph in enumerate(phases):
    f = sum(1 for op, l in ph if cls(op, l) == "fast"); s = sum(1 for op, l in ph if cls(op, l) == "slow")
    sa = sum(1 for op, l in ph if op.startswith("s_") and op not in ("s_nop", "s_waitcnt")); ld = sum(1 for op, l in ph if op.startswith("ds_"))
    tot.append((f, s))
    print("%-20s %6d %6d %6d %6d %6d  %.2f" % (names[i] if i < len(names) else "?", f + s, f, s, sa, ld, f / max(f + s, 1)))
if len(sys.argv) >= 6:
    ex = [float(x) for x in sys.argv[2:6]]          # executed VALU per wave: P1, P2, P3, P4 (differences of the ablation levels)
    groups = {"P1 colour": [1], "P2 fDCT + quantize": [2], "P3 entropy": [3, 4], "P4 scan, stitch, flush": [5, 6, 7]}
    print("\n%-26s %9s %10s %14s %9s" % ("phase (executed, per wave)", "VALU", "fast share", "SIMD cycles", "share"))
    rows = []
    for (nm, idx), n in zip(groups.items(), ex):
        f = sum(tot[i][0] for i in idx if i < len(tot)); s = sum(tot[i][1] for i in idx if i < len(tot))
        fs = f / max(f + s, 1)
        rows.append((nm, n, fs, n * (fs * C_FAST + (1 - fs) * C_SLOW)))
    allc = sum(r[3] for r in rows)
    for nm, n, fs, c in rows:
        print("%-26s %9.0f %10.2f %14.0f %9.3f" % (nm, n, fs, c, c / allc))
    print("%-26s %9.0f %10s %14.0f   (x 202 500 waves / 1 024 SIMDs / 2.4 GHz = %.3f ms)" % ("all", sum(ex), "", allc, allc * 202500 / 1024 / 2.4e6))
