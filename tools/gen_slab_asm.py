#!/usr/bin/env python3
"""Generates ntpoly_amd/csrc/slab_loop.inc: the hand-scheduled main loop of k_spgemm_slab as ONE inline-asm
block (gfx950).  Every register the loop touches is a fixed physical register, so nothing the compiler does
can read an SGPR/VGPR whose asynchronous load has not landed.

Pipeline (step t = column k = kmin + t of A):
  top of step t : s_waitcnt lgkmcnt(0)      -> B(k, :) of step t and the run record of step t+2 have landed
                  issue the SL slab loads of step t+2 (buffer_load, zero fill outside the run)
                  issue s_load of B(k+1, :) and of the run record of step t+3
                  s_waitcnt vmcnt(2*SL)     -> the slabs of step t have landed (t+1, t+2 still in flight)
                  for every slab the run touches: 16 x (v_mul_f64, v_add_f64), 4 independent chains at a time
Register map (J = 16 columns, SL = 3 slabs per wave):
  s[14:15] run-record pointer (advances 32 B per step)   s[16:17] multiplier pointer (128 B per step)
  s18 step counter   s19 scratch   (first, span62) of the step on slab set 0/1/2: (s20,s21) (s22,s23) (s12,s13)
  s[24:31] run record: buffer descriptor s[24:27], first8 s28, first s29, span62 s30
  s[36:67] multipliers B(k, 0..15) of set 0          s[68:99] of set 1
  (s32..s35 are left alone: s32 is the ABI stack pointer)
  v[2:33], v[34:65], v[66:97]  accumulators of slab 0, 1, 2 (16 doubles each)
  v[98:105] products in flight   v[106:108] load offsets
  slab values: set 0 v[116:121], set 1 v[122:127], set 2 v[110:115]
"""
import os

SL, J = 3, 16
ACC0 = 2          # first accumulator VGPR
T0 = 98
VOFF = 106
A_SET = [116, 122, 110]
B_SET = [36, 68]
FS = [(20, 21), (22, 23), (12, 13)]
NA, NB = len(A_SET), len(B_SET)
PF = 4           # scalar-cache prefetch distance of the multiplier rows, in steps (0 = off)


def vp(n):
    return "v[%d:%d]" % (n, n + 1)


def sp(n):
    return "s[%d:%d]" % (n, n + 1)


RO = [100, 101]  # row-offset variant: byte offset of the multiplier row of step k (from the pad word of its record)


def issue(aset, lines, ablate=0, ro=None):
    """copy first/span62 of the record that just landed, issue the SL slab loads of that step"""
    f, s = FS[aset]
    lines.append("s_mov_b32 s%d, s29" % f)
    lines.append("s_mov_b32 s%d, s30" % s)
    if ro is not None:
        lines.append("s_mov_b32 s%d, s31" % ro)
    lines.append("v_sub_u32 v%d, %%[r0], s28" % VOFF)
    for i in range(1, SL):
        lines.append("v_add_u32 v%d, %%[c%d], v%d" % (VOFF + i, i, VOFF))
    for i in range(SL):
        if ablate == 1:
            continue
        lines.append("buffer_load_dwordx2 %s, v%d, s[24:27], 0 offen" % (vp(A_SET[aset] + 2 * i), VOFF + i))


def load_b(bset, off, lines, ablate=0, ro=None):
    if ablate == 2:
        return
    if ro is not None:   # the row of the step is named by its record, not by its position
        lines.append("s_load_dwordx16 s[%d:%d], s[16:17], s%d" % (B_SET[bset], B_SET[bset] + 15, ro))
        lines.append("s_load_dwordx16 s[%d:%d], s[16:17], s%d offset:0x40" % (B_SET[bset] + 16, B_SET[bset] + 31, ro))
        return
    lines.append("s_load_dwordx16 s[%d:%d], s[16:17], 0x%x" % (B_SET[bset], B_SET[bset] + 15, off))
    lines.append("s_load_dwordx16 s[%d:%d], s[16:17], 0x%x" % (B_SET[bset] + 16, B_SET[bset] + 31, off + 0x40))


def compute(aset, bset, lines, label, fused, ablate=0):
    f, s = FS[aset]
    for i in range(SL):
        skip = "%d" % label[0]
        label[0] += 1
        lines.append("s_sub_i32 s19, %%[e%d], s%d" % (i, f))
        lines.append("s_cmp_gt_u32 s19, s%d" % s)
        lines.append("s_cbranch_scc1 %sf" % skip)
        a = vp(A_SET[aset] + 2 * i)
        for g in range(J // 4 if ablate != 3 else 0):
            if fused:
                for q in range(4):
                    acc = vp(ACC0 + 2 * J * i + 2 * (4 * g + q))
                    lines.append("v_fma_f64 %s, %s, %s, %s" % (acc, a, sp(B_SET[bset] + 2 * (4 * g + q)), acc))
                continue
            for q in range(4):
                lines.append("v_mul_f64 %s, %s, %s" % (vp(T0 + 2 * q), a, sp(B_SET[bset] + 2 * (4 * g + q))))
            for q in range(4):
                acc = vp(ACC0 + 2 * J * i + 2 * (4 * g + q))
                lines.append("v_add_f64 %s, %s, %s" % (acc, acc, vp(T0 + 2 * q)))
        lines.append("%s:" % skip)


def step(L, label, t, fused, ablate, pf, lean, rowoff=False):
    """one k step at position t of the period.  lean: the pointers stay at the period start (immediate offsets carry
    the step) and there is no exit test -- the caller guarantees a whole period.  rowoff: the multiplier row of a step
    sits at the byte offset its run record names (pad word), s[16:17] stays at the tile"""
    L.append("s_waitcnt lgkmcnt(0)")
    issue((t + 2) % NA, L, ablate, RO[t % 2] if rowoff else None)      # (step t+2; RO[t%2] held step t's, consumed)
    if lean:
        boff, roff = 0x80 * (t + 1), 0x20 * (t + 3)
    else:
        # pointers advance once per step: s[16:17] -> multipliers of step t+1, s[14:15] -> record of step t+1
        if not rowoff:
            L.append("s_add_u32 s16, s16, 0x80")
            L.append("s_addc_u32 s17, s17, 0")
        L.append("s_add_u32 s14, s14, 0x20")
        L.append("s_addc_u32 s15, s15, 0")
        boff, roff = 0, 0x40
    load_b((t + 1) % NB, boff, L, ablate, RO[(t + 1) % 2] if rowoff else None)
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x%x" % roff)   # record of step t+3
    if rowoff and pf:
        # (row-offset variant) the row of step t+2 -- its offset has just arrived with its record -- is pulled into the
        # scalar cache by one wave of the block per step, the waves taking turns as below
        skip = "%d" % label[0]
        label[0] += 1
        if lean:
            L.append("s_add_i32 s19, s18, %d" % t)
            L.append("s_and_b32 s19, s19, 3")
        else:
            L.append("s_and_b32 s19, s18, 3")
        L.append("s_cmp_lg_u32 s19, %[wv]")
        L.append("s_cbranch_scc1 %sf" % skip)
        L.append("s_load_dword vcc_lo, s[16:17], s%d" % RO[t % 2])   # (destination: a register the loop does not use)
        L.append("s_load_dword vcc_lo, s[16:17], s%d offset:0x40" % RO[t % 2])
        L.append("%s:" % skip)
    elif pf:
        # scalar-cache prefetch of the multiplier row PF steps further on, issued by ONE wave of the block per
        # step (the wave whose index equals step mod 4): s_waitcnt lgkmcnt(0) makes the issuing wave sit out the
        # full miss latency at its next step, so the four waves take turns and each stalls once in four steps
        # while the other three find their rows in the cache.  s31 is the unused pad word of the record.
        skip = "%d" % label[0]
        label[0] += 1
        if lean:   # s18 counts whole periods here: the step within the period is known at generation time
            L.append("s_add_i32 s19, s18, %d" % t)
            L.append("s_and_b32 s19, s19, 3")
            poff = 0x80 * (t + pf)
        else:
            L.append("s_and_b32 s19, s18, 3")
            poff = pf * 0x80
        L.append("s_cmp_lg_u32 s19, %[wv]")
        L.append("s_cbranch_scc1 %sf" % skip)
        L.append("s_load_dword s31, s[16:17], 0x%x" % poff)
        L.append("s_load_dword s31, s[16:17], 0x%x" % (poff + 0x40))
        L.append("%s:" % skip)
    L.append("s_waitcnt vmcnt(%d)" % (2 * SL))
    compute(t % NA, t % NB, L, label, fused, ablate)


def build(fused, ablate=0, pf=0, lean=False, rowoff=False):
    L = []
    label = [10]
    for i in range(SL * J):
        L.append("v_mov_b64 %s, 0" % vp(ACC0 + 2 * i))
    L.append("s_mov_b64 s[14:15], %[rp]")
    L.append("s_mov_b64 s[16:17], %[bq]")
    # prologue: slabs of steps 0 and 1, multipliers of step 0, record of step 2
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x0")
    L.append("s_waitcnt lgkmcnt(0)")
    issue(0, L, ablate, RO[0] if rowoff else None)
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x20")
    L.append("s_waitcnt lgkmcnt(0)")
    issue(1, L, ablate, RO[1] if rowoff else None)
    load_b(0, 0, L, ablate, RO[0] if rowoff else None)
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x40")
    L.append("s_mov_b32 s18, 0")
    period = NA * NB
    if lean:
        # whole periods first: no pointer arithmetic and no exit test inside (7 scalar instructions less per step)
        L.append("3:")
        L.append("s_sub_i32 s19, %[kn], s18")
        L.append("s_cmp_lt_i32 s19, %d" % period)
        L.append("s_cbranch_scc1 4f")
        for t in range(period):
            step(L, label, t, fused, ablate, pf, True, rowoff)
        if not rowoff:
            L.append("s_add_u32 s16, s16, 0x%x" % (0x80 * period))
            L.append("s_addc_u32 s17, s17, 0")
        L.append("s_add_u32 s14, s14, 0x%x" % (0x20 * period))
        L.append("s_addc_u32 s15, s15, 0")
        L.append("s_add_i32 s18, s18, %d" % period)
        L.append("s_branch 3b")
        L.append("4:")
        L.append("s_cmp_ge_i32 s18, %[kn]")
        L.append("s_cbranch_scc1 2f")
    L.append("1:")
    for t in range(period):
        step(L, label, t, fused, ablate, pf, False, rowoff)
        L.append("s_add_i32 s18, s18, 1")
        L.append("s_cmp_ge_i32 s18, %[kn]")
        if t + 1 < period:
            L.append("s_cbranch_scc1 2f")
        else:
            L.append("s_cbranch_scc0 1b")
    L.append("2:")
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


# ------------------------------------------------------------------------------------------------- complex operands
# Register map of the complex loop (J = 8 complex columns, SL = 2 slabs per wave, NW = 6 waves):
#   SGPRs as above; a multiplier set holds 8 complex numbers (re, im pairs): B(k, c) = s[base+4c : base+4c+3]
#   v[2:33], v[34:65]  accumulators of slab 0, 1: column c -> (re, im) = v[.. + 4c : .. + 4c + 3]
#   v[66:73] four temporaries   v[74:75] load offsets   slab values (re, im per lane): set 0 v[76:83], set 1 v[84:91]
C_SL, C_J = 2, 8
C_ACC0, C_T0, C_VOFF = 2, 66, 74
C_A_SET = [76, 84]


def c_issue(aset, lines):
    f, s = FS[aset]
    lines.append("s_mov_b32 s%d, s29" % f)
    lines.append("s_mov_b32 s%d, s30" % s)
    lines.append("v_sub_u32 v%d, %%[r0], s28" % C_VOFF)
    lines.append("v_add_u32 v%d, %%[c1], v%d" % (C_VOFF + 1, C_VOFF))
    for i in range(C_SL):
        a = C_A_SET[aset] + 4 * i
        lines.append("buffer_load_dwordx4 v[%d:%d], v%d, s[24:27], 0 offen" % (a, a + 3, C_VOFF + i))


def c_compute(aset, bset, lines, label):
    """(ar + i ai)(br + i bi): four products, one subtraction, one addition, then the two accumulates -- every operation
    rounded on its own, as the reference's complex multiply-add.  One column at a time on four temporaries: with 92
    VGPRs five waves fit a SIMD (the loop is latency bound, so residency beats interleaving two columns on sixteen)."""
    f, s = FS[aset]
    for i in range(C_SL):
        skip = "%d" % label[0]
        label[0] += 1
        lines.append("s_sub_i32 s19, %%[e%d], s%d" % (i, f))
        lines.append("s_cmp_gt_u32 s19, s%d" % s)
        lines.append("s_cbranch_scc1 %sf" % skip)
        ar, ai = vp(C_A_SET[aset] + 4 * i), vp(C_A_SET[aset] + 4 * i + 2)
        t = [vp(C_T0 + 2 * q) for q in range(4)]
        for c in range(C_J):
            br, bi = sp(B_SET[bset] + 4 * c), sp(B_SET[bset] + 4 * c + 2)
            lines.append("v_mul_f64 %s, %s, %s" % (t[0], ar, br))
            lines.append("v_mul_f64 %s, %s, %s" % (t[1], ai, bi))
            lines.append("v_mul_f64 %s, %s, %s" % (t[2], ar, bi))
            lines.append("v_mul_f64 %s, %s, %s" % (t[3], ai, br))
            lines.append("v_add_f64 %s, %s, -%s" % (t[0], t[0], t[1]))
            lines.append("v_add_f64 %s, %s, %s" % (t[2], t[2], t[3]))
            accr = vp(C_ACC0 + 4 * C_J * i + 4 * c)
            acci = vp(C_ACC0 + 4 * C_J * i + 4 * c + 2)
            lines.append("v_add_f64 %s, %s, %s" % (accr, accr, t[0]))
            lines.append("v_add_f64 %s, %s, %s" % (acci, acci, t[2]))
        lines.append("%s:" % skip)


def build_complex():
    L = []
    label = [10]
    for i in range(C_SL * C_J * 2):
        L.append("v_mov_b64 %s, 0" % vp(C_ACC0 + 2 * i))
    L.append("s_mov_b64 s[14:15], %[rp]")
    L.append("s_mov_b64 s[16:17], %[bq]")
    # prologue: slabs of step 0, multipliers of step 0, record of step 1
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x0")
    L.append("s_waitcnt lgkmcnt(0)")
    c_issue(0, L)
    load_b(0, 0, L)
    L.append("s_load_dwordx8 s[24:31], s[14:15], 0x20")
    L.append("s_mov_b32 s18, 0")
    L.append("1:")
    for t in range(2):
        L.append("s_waitcnt lgkmcnt(0)")       # B(k, :) of step t and the record of step t+1
        c_issue((t + 1) % 2, L)
        L.append("s_add_u32 s16, s16, 0x80")
        L.append("s_addc_u32 s17, s17, 0")
        L.append("s_add_u32 s14, s14, 0x20")
        L.append("s_addc_u32 s15, s15, 0")
        load_b((t + 1) % 2, 0, L)
        L.append("s_load_dwordx8 s[24:31], s[14:15], 0x20")   # record of step t+2
        L.append("s_waitcnt vmcnt(%d)" % C_SL)
        c_compute(t % 2, t % 2, L, label)
        L.append("s_add_i32 s18, s18, 1")
        L.append("s_cmp_ge_i32 s18, %[kn]")
        if t == 0:
            L.append("s_cbranch_scc1 2f")
        else:
            L.append("s_cbranch_scc0 1b")
    L.append("2:")
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


def geometry(sl):
    """register map for SL slabs per wave (the accumulators start at v2; everything else follows them):
    SL = 3: the map in the header (128 VGPRs, 4 waves per SIMD); SL = 2: 88 VGPRs (5 waves); SL = 1: 50 VGPRs (8 waves)"""
    global SL, T0, VOFF, A_SET
    SL = sl
    if sl == 3:
        T0, VOFF, A_SET = 98, 106, [116, 122, 110]
    else:
        T0 = ACC0 + 2 * J * sl
        VOFF = T0 + 8
        a0 = VOFF + sl + (VOFF + sl) % 2          # slab values sit on even registers
        A_SET = [a0, a0 + 2 * sl, a0 + 4 * sl]
    return T0, VOFF, A_SET


def main():
    sclob = ["s%d" % i for i in [12, 13] + list(range(14, 32)) + list(range(36, 100))]
    vclob = ["v%d" % i for i in list(range(T0, VOFF + SL)) + list(range(110, 128))]
    out = []
    out.append("// GENERATED by tools/gen_slab_asm.py -- do not edit.  Main loop of k_spgemm_slab<16,3,NW>.")
    out.append("// Operands: outputs accL0,accH0,accL1,accH1,accL2,accH2 (v8d, pinned to v[2:97]); inputs rp, bq (64-bit")
    out.append("// uniform pointers), kn, e0..e2 (last row of the wave's slabs), r0 (per-lane row offset * 8 of slab 0), c1, c2")
    out.append("// (immediates: byte distance of slabs 1 and 2 from slab 0), wv (index of the wave in its workgroup, 0..3).")
    # the rotating prefetch pays off only when the loop is not ALU bound: measured 2.69 -> 2.30 ms with v_fma_f64,
    # 2.81 -> 2.85 ms with separate multiply and add (kept selectable as variant 405)
    for name, fused, abl, pf, lean in (("SLAB_LOOP_ASM", False, 0, 0, False), ("SLAB_LOOP_ASM_FMA", True, 0, PF, False),
                                       ("SLAB_LOOP_ASM_ABL1", False, 1, 0, False), ("SLAB_LOOP_ASM_ABL2", False, 2, 0, False),
                                       ("SLAB_LOOP_ASM_ABL3", False, 3, 0, False), ("SLAB_LOOP_ASM_PF", False, 0, PF, False),
                                       ("SLAB_LOOP_ASM_LEAN", False, 0, 0, True), ("SLAB_LOOP_ASM_LEANPF", False, 0, PF, True)):
        L = build(fused, abl, pf, lean)
        out.append("#define %s \\" % name)
        for ln in L:
            out.append('  "%s\\n\\t" \\' % ln)
        out.append('  ""')
        print(name, len(L), "instructions")
    out.append("#define SLAB_LOOP_CLOBBERS " + ", ".join('"%s"' % c for c in sclob + vclob) + ', "vcc", "scc", "memory"')
    # label-ordered steps: the multiplier row of a step at the byte offset its record names (lean periods, no rotating
    # prefetch: the rows are not consecutive); two more SGPRs carry the offsets of the steps in flight
    L = build(False, 0, PF, True, True)
    out.append("#define SLAB_LOOP_ASM_ROWOFF \\")
    for ln in L:
        out.append('  "%s\\n\\t" \\' % ln)
    out.append('  ""')
    print("SLAB_LOOP_ASM_ROWOFF", len(L), "instructions")
    out.append("#define SLAB_LOOP_ROWOFF_CLOBBERS " + ", ".join('"%s"' % c for c in sclob + ["s100", "s101"] + vclob) +
               ', "vcc", "scc", "memory"')
    # narrower row windows: fewer slabs per wave = fewer accumulator registers = more waves per SIMD (the loop is
    # latency bound: time ~ 1 / occupancy, profiles/README.md item 14)
    for sl in (1, 2):
        t0, voff, a_set = geometry(sl)
        L = build(False)
        out.append("#define SLAB_LOOP_ASM_SL%d \\" % sl)
        for ln in L:
            out.append('  "%s\\n\\t" \\' % ln)
        out.append('  ""')
        vc = ["v%d" % i for i in range(t0, a_set[2] + 2 * sl)]
        out.append("#define SLAB_LOOP_SL%d_CLOBBERS " % sl + ", ".join('"%s"' % c for c in sclob + vc) + ', "vcc", "scc", "memory"')
        print("SLAB_LOOP_ASM_SL%d" % sl, len(L), "instructions; VGPRs up to v%d" % (a_set[2] + 2 * sl - 1))
    geometry(3)
    # complex loop: outputs accA..accD (v8d, v[2:65]); inputs rp, bq, kn, e0, e1, r0 (row offset * 16), c1 (immediate)
    L = build_complex()
    out.append("#define SLAB_LOOP_ASM_CPLX \\")
    for ln in L:
        out.append('  "%s\\n\\t" \\' % ln)
    out.append('  ""')
    print("SLAB_LOOP_ASM_CPLX", len(L), "instructions")
    cs = ["s%d" % i for i in [12, 13] + list(range(14, 32)) + list(range(36, 100))]
    cv = ["v%d" % i for i in range(C_T0, C_A_SET[1] + 4 * C_SL)]
    out.append("#define SLAB_LOOP_CPLX_CLOBBERS " + ", ".join('"%s"' % c for c in cs + cv) + ', "vcc", "scc", "memory"')
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ntpoly_amd", "csrc", "slab_loop.inc")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
