#!/usr/bin/env python3
"""Generator of the instruction stream of ``attn_bwd2_dkdv_w64_kernel`` (attention_bwd2.hip): the dK / dV half of the
flash-attention backward (head_dim 128), one wave per SIMD with an asm-owned register map.

    python gen_attn_bwd_w64.py > attention_bwd2_asm.inc          (build.py does this when the generator changed)

Same mathematics, tiles and LDS layout as the HIP kernel it replaces (attention_bwd2.hip documents them): a workgroup
is 4 waves x 32 keys, the K / V fragments of a wave's keys stay in registers, the loop runs over tiles of 64 queries
whose Q and dO rows are LDS-DMA staged ([64][128] bf16, 16-byte slots XOR-swizzled), and per 32-query half

    A   S = Q K^T, dP = dO V^T          16 MFMA  (operand A: row fragments, ds_read_b128)
    B   P = exp2(S), dS = P dP           48 VALU  (S starts from -lse', dP from -delta: MFMA C operand)
    C   dV^T += dO^T P, dK^T += Q^T dS   16 MFMA  (operand A: transposed fragments, 2 x ds_read_b64_tr_b16)

What the compiler could not be made to do (round 5 measurements, profiles/r05_attn_bwd_w64.json): its schedule issued
the fragment reads of a stage, waited, ran the 16 MFMAs, then all the VALU work — and 192 v_accvgpr moves per tile —
5 300 cycles per tile for 2 048 cycles of matrix work.  Here one tile is ONE software-pipelined sequence of 64 MFMAs,

    A0 | A1 + B0 | C0 + B1 | C1        (B rides in the gaps of the next 16 MFMAs)

every fragment is requested 6-12 MFMAs ahead into an 8-stage register ring (across stage and tile boundaries), the
tile ring in LDS has FOUR slots filled two tiles ahead (one barrier per tile, placed where the next tile's first
fragments are requested; `s_waitcnt vmcnt(10)` leaves the newest tile in flight), and lse / delta reach the lanes
through LDS: raw values by LDS-DMA, one fix-up per query ((lse > -inf) ? -lse log2(e) / sc : -inf, and -delta; every wave
does it redundantly — identical values, no barrier), then 8 ds_read_b128 per half straight into the accumulators.

Register map (asm-owned; the per-lane inputs stay in the compiler's v0..v15):
  a[0:63]    dV^T accumulators [db] 16 each        a[64:127]  dK^T accumulators
  a[128:159] K fragments [kk] 4 each               a[160:191] V fragments
  v[16:23]   row-fragment LDS addresses (per kk)   v[24:27] / v[28:31]  transposed-fragment addresses (first / second read)
  v32 final-statistics read address   v33 raw-statistics address (this lane's query)   v[36:42] temporaries
  v[64:79] S half 0   v[80:95] dP half 0   v[96:111] S half 1   v[112:127] dP half 1
  v[128:159] packed P / dS  [half][P | dS][a] 4 each
  v[192:255] fragment ring: 8 stages x (first operand 4 | second operand 4)
  s[80:95]   slot offsets of the four address groups, their steps, loop counter, DMA source offset
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_w64 import Emit, spread, vr, ar             # noqa: E402

SLOT = 32768                      # one ring slot: Q tile 16 KiB | dO tile 16 KiB
NSLOT = 4
STAT_RAW = NSLOT * SLOT           # raw statistics: 4 slots x [lse 64 | delta 64] fp32
STAT_FIN = STAT_RAW + NSLOT * 512 # fixed-up statistics, same shape
LDS_BYTES = STAT_FIN + NSLOT * 512
PARK_BLOCK = 1040                 # the epilogue parks 4 accumulator registers x 64 lanes per block; 16 bytes of padding: the
PARK_WAVE = 32 * PARK_BLOCK       # row-major read-back (16 lanes = 16 blocks of one row) then hits 64 different banks
ABL = os.environ.get("OMH_BWD_ABL", "").split(",")     # timing-only ablations (wrong numerics by construction; never shipped)
LEAD_A, LEAD_C = 12, 6            # a fragment is requested this many MFMAs ahead (stage A: one read each; C: two) — the
                                  # LDS counter holds 15 outstanding operations


# ---------------------------------------------------------------- register map
def DV(db):        return db * 16
def DK(db):        return 64 + db * 16
def KF(kk):        return 128 + kk * 4
def VF(kk):        return 160 + kk * 4
ROW = list(range(16, 24))
TRA = list(range(24, 28))
TRB = list(range(28, 32))
STR, RAW = 32, 33
T0, T1 = 36, 37
NEGINF = 42
def S(hb):         return 64 + hb * 32
def DP(hb):        return 80 + hb * 32
def PK(hb, ds, a): return 128 + hb * 16 + ds * 8 + a * 4
def RING(st, w):   return 192 + st * 8 + w * 4

S_ROW, S_TR, S_ST, S_DMA = "s80", "s81", "s82", "s83"           # slot byte offsets of the four address groups
D_ROW, D_TR, D_ST = "s84", "s85", "s86"                          # the step each group's addresses take next
C_STEP, C_WRAP, C_STSTEP, C_STWRAP = "s87", "s88", "s89", "s90"  # 32768, -3*32768, 512, -3*512
S_CNT, S_OFF = "s91", "s92"


def advance(sreg, dreg, step_c, wrap_c, nbytes):
    return [f"s_add_u32 {sreg}, {sreg}, {step_c}", f"s_cmp_eq_u32 {sreg}, {nbytes}", f"s_cselect_b32 {dreg}, {wrap_c}, {step_c}",
            f"s_cselect_b32 {sreg}, 0, {sreg}"]


# ---------------------------------------------------------------- ops and the LDS wait tracker
# op = (kind, text, needs, tag, prep)   kind "m" MFMA | "r" ds_read (tag) | "x" anything else; needs: read tags that must have
# landed; prep: belongs to the NEXT tile's preparation (the prologue replays exactly these to enter the loop)
def op(kind, text, needs=(), tag=None, prep=False):
    return (kind, text, tuple(needs), tag, prep)


def linearize(e, ops, pending):
    pending = list(pending)
    for kind, text, needs, tag, _ in ops:
        need = [t for t in needs if t in pending]
        if need:
            last = max(pending.index(t) for t in need)
            allowed = min(len(pending) - last - 1, 15)           # (the counter has 4 bits: more than 15 cannot be outstanding)
            e(f"s_waitcnt lgkmcnt({allowed})")
            pending = pending[len(pending) - allowed:] if allowed else []
        e(text)
        if kind == "r":
            pending.append(tag)
        elif kind == "w":                                       # ds_write: counts in lgkmcnt too
            pending.append(tag)
    return pending


def mfma_of(i):
    """MFMA i of a tile (0..63) -> (text, needs)."""
    pr, w = i // 2, i % 2
    st = pr % 8
    frag = vr(RING(st, w), 4)
    if i < 32:
        hb, kk = i // 16, (i % 16) // 2
        acc = vr((S(hb) if w == 0 else DP(hb)), 16)
        b = ar((KF(kk) if w == 0 else VF(kk)), 4)
        needs = [] if "norow" in ABL else [f"F{i}"]
        if kk == 0 and "nostat" not in ABL:
            needs += [f"ST{hb}.{w}.{g}" for g in range(4)]
        return f"v_mfma_f32_32x32x16_bf16 {acc}, {frag}, {b}, {acc}", needs
    j = i - 32
    hb, a, db = j // 16, (j % 16) // 8, (j % 8) // 2
    acc = ar((DV(db) if w == 0 else DK(db)), 16)
    b = vr(PK(hb, w, a), 4)
    return f"v_mfma_f32_32x32x16_bf16 {acc}, {frag}, {b}, {acc}", ([] if "notr" in ABL else [f"F{i}a", f"F{i}b"])


def frag_reads(i, prep):
    """The LDS reads that feed MFMA i (issued LEAD_A / LEAD_C MFMAs earlier)."""
    pr, w = i // 2, i % 2
    st = pr % 8
    if ("norow" in ABL and i < 32) or ("notr" in ABL and i >= 32):
        return []
    if i < 32:
        hb, kk = i // 16, (i % 16) // 2
        off = hb * 8192 + (16384 if w == 1 else 0)              # S: Q rows;  dP: dO rows
        return [op("r", f"ds_read_b128 {vr(RING(st, w), 4)}, {vr(ROW[kk])} offset:{off}", tag=f"F{i}", prep=prep)]
    j = i - 32
    hb, a, db = j // 16, (j % 16) // 8, (j % 8) // 2
    off = hb * 8192 + a * 4096 + (16384 if w == 0 else 0)       # dV: dO^T;  dK: Q^T
    return [op("r", f"ds_read_b64_tr_b16 {vr(RING(st, w), 2)}, {vr(TRA[db])} offset:{off}", tag=f"F{i}a", prep=prep),
            op("r", f"ds_read_b64_tr_b16 {vr(RING(st, w) + 2, 2)}, {vr(TRB[db])} offset:{off}", tag=f"F{i}b", prep=prep)]


def stage_b(hb, pre):
    """P = exp2(S) (in place), dS = P dP (in place), both packed to bf16 pairs: the MFMA B operands of stage C."""
    out = []
    for ee in range(8):
        r0, r1 = 2 * ee, 2 * ee + 1
        if not pre:
            out += [f"v_mul_f32 {vr(S(hb) + r0)}, %[sc], {vr(S(hb) + r0)}", f"v_mul_f32 {vr(S(hb) + r1)}, %[sc], {vr(S(hb) + r1)}"]
        out += [f"v_exp_f32 {vr(S(hb) + r0)}, {vr(S(hb) + r0)}",
                f"v_exp_f32 {vr(S(hb) + r1)}, {vr(S(hb) + r1)}",
                f"v_mul_f32 {vr(DP(hb) + r0)}, {vr(S(hb) + r0)}, {vr(DP(hb) + r0)}",     # (the exp2 two instructions back: trans hazard kept)
                f"v_mul_f32 {vr(DP(hb) + r1)}, {vr(S(hb) + r1)}, {vr(DP(hb) + r1)}",
                f"v_cvt_pk_bf16_f32 {vr(PK(hb, 0, ee // 4) + ee % 4)}, {vr(S(hb) + r0)}, {vr(S(hb) + r1)}",
                f"v_cvt_pk_bf16_f32 {vr(PK(hb, 1, ee // 4) + ee % 4)}, {vr(DP(hb) + r0)}, {vr(DP(hb) + r1)}"]
    return [op("x", ln) for ln in out]


def stat_fixup():
    """Raw lse / delta of this lane's query (tile t+1) -> fixed-up values in the `final` area of the same slot."""
    return [op("r", f"ds_read_b32 {vr(T0)}, {vr(RAW)}", tag="RAWL", prep=True),
            op("r", f"ds_read_b32 {vr(T1)}, {vr(RAW)} offset:256", tag="RAWD", prep=True),
            op("x", f"v_mul_f32 {vr(T0)}, %[k1], {vr(T0)}", needs=["RAWL", "RAWD"], prep=True),      # -lse log2(e) / sc
            op("x", f"v_xor_b32 {vr(T1)}, 0x80000000, {vr(T1)}", prep=True),                          # -delta
            op("x", f"v_cmp_eq_u32 vcc, 0x7f800000, {vr(T0)}", prep=True),                            # lse = -inf: no keys, P = 0
            op("x", f"v_cndmask_b32 {vr(T0)}, {vr(T0)}, {vr(NEGINF)}, vcc", prep=True),
            op("w", f"ds_write_b32 {vr(RAW)}, {vr(T0)} offset:{STAT_FIN - STAT_RAW}", tag="FINL", prep=True),
            op("w", f"ds_write_b32 {vr(RAW)}, {vr(T1)} offset:{STAT_FIN - STAT_RAW + 256}", tag="FIND", prep=True)]


def stat_reads(hb):
    """S / dP of half hb start from the statistics of their 16 queries: register r <-> query 32 hb + 16 (r>>3) + 8 lh + (r&7)."""
    out = []
    for w in range(2):
        for g in range(4):
            a, h = g // 2, g % 2
            off = (32 * hb + 16 * a + 4 * h) * 4 + 256 * w
            dst = (S(hb) if w == 0 else DP(hb)) + 4 * g
            out.append(op("r", f"ds_read_b128 {vr(dst, 4)}, {vr(STR)} offset:{off}", tag=f"ST{hb}.{w}.{g}", prep=True))
    return out


def dma_group(slot_expr):
    """One tile's group of 10 loads: lse and delta of its 64 queries (one dword per lane each) + the 8 pieces of its Q and dO
    rows -> ring slot."""
    out = [f"s_lshr_b32 {S_OFF}, {slot_expr}, 6" if not isinstance(slot_expr, int) else f"s_mov_b32 {S_OFF}, {slot_expr >> 6}",
           f"s_add_u32 m0, %[sraw], {S_OFF}",
           "s_nop 0",
           "buffer_load_dword %[vost], %[rlse], %[sstn] offen lds",
           "s_add_u32 m0, m0, 256",
           "s_nop 0",
           "buffer_load_dword %[vost], %[rdel], %[sstn] offen lds",
           "s_add_u32 %[sstn], %[sstn], 256"]
    for which, (vo, rs, sn, sp) in enumerate((("%[vodq]", "%[rq]", "%[sqn]", "%[sqp]"), ("%[vodo]", "%[rdo]", "%[son]", "%[sop]"))):
        for j in range(4):
            if j == 0:
                out.append(f"s_add_u32 m0, %[ldsw], {slot_expr}" if which == 0 else "s_add_u32 m0, m0, 0x1000")
                out.append(f"s_mov_b32 {S_OFF}, {sn}")
            else:
                out.append("s_add_u32 m0, m0, 0x1000")
                out.append(f"s_add_u32 {S_OFF}, {S_OFF}, {sp}")
            out.append(f"buffer_load_dwordx4 {vo}, {rs}, {S_OFF} offen lds")
        out.append(f"s_add_u32 {sn}, {S_OFF}, {sp}")                # next tile
    return out


def tile_ops(pre):
    """The 64 gaps of one tile: gaps[i] = ops issued right after MFMA i."""
    gaps = [[] for _ in range(64)]
    # fragment reads, LEAD MFMAs ahead (the last gaps feed the next tile's stage A0)
    for tgt in range(64, 128):
        lead = LEAD_A if tgt % 64 < 32 else LEAD_C
        g = tgt - lead
        gaps[g % 64] += frag_reads(tgt % 64, prep=g < 64)
    for tgt in range(64):                                            # ring stage re-use: the previous user (pair - 8) has been issued
        lead = LEAD_A if tgt < 32 else LEAD_C
        assert 2 * (tgt // 2 - 8) + 1 <= tgt - lead, tgt
    # stage B, half of it at a time: pairs 0..3 make the a = 0 operands (first read by the 1st MFMA of stage C), pairs 4..7
    # the a = 1 operands (first read by the 9th) — spread as widely as those deadlines allow: ~2 VALU per gap
    b0, b1 = stage_b(0, pre), stage_b(1, pre)
    if "nob" in ABL:
        b0, b1 = [], []
    for lst, lo, hi in ((b0[:len(b0) // 2], 17, 30), (b0[len(b0) // 2:], 30, 38), (b1[:len(b1) // 2], 34, 46), (b1[len(b1) // 2:], 46, 54)):
        for g, ops_ in enumerate(spread(lst, 64, lo, hi)):
            gaps[g] += ops_
    # address groups: row addresses -> next slot once the last row read of this tile is out (gap 19), the transposed
    # ones after gap 51, the statistics ones right before their use
    # (the four scalar instructions of an advance stay together: s_cmp -> s_cselect must not see a foreign SCC write)
    gaps[20] += [op("x", ln) for ln in advance(S_ROW, D_ROW, C_STEP, C_WRAP, NSLOT * SLOT)]
    for g, lst in enumerate(spread([op("x", f"v_add_u32 {vr(r)}, {D_ROW}, {vr(r)}") for r in ROW], 64, 21, 33)):
        gaps[g] += lst
    gaps[60] += [op("x", ln) for ln in advance(S_TR, D_TR, C_STEP, C_WRAP, NSLOT * SLOT)]
    for g, lst in enumerate(spread([op("x", f"v_add_u32 {vr(r)}, {D_TR}, {vr(r)}") for r in TRA + TRB], 64, 61, 64)):
        gaps[g] += lst
    st_adv = [op("x", ln) for ln in advance(S_ST, D_ST, C_STSTEP, C_STWRAP, NSLOT * 512)] + \
             [op("x", f"v_add_u32 {vr(STR)}, {D_ST}, {vr(STR)}"), op("x", f"v_add_u32 {vr(RAW)}, {D_ST}, {vr(RAW)}")]
    gaps[45] += st_adv
    # the barrier of the tile: everything of tile t+1 has landed (the 10 loads of tile t+2 may still fly) and every wave is
    # done with tile t-1, whose slot the loads of tile t+3 now overwrite
    gaps[47] += [op("x", "s_waitcnt vmcnt(10)"), op("x", "s_barrier")]
    gaps[48] += [op("x", ln) for ln in advance(S_DMA, S_OFF, C_STEP, C_WRAP, NSLOT * SLOT)[:2] + [f"s_cselect_b32 {S_DMA}, 0, {S_DMA}"]]
    for g, lst in enumerate(spread([op("x", ln) for ln in dma_group(S_DMA) if not ("nodma" in ABL and ln.startswith("buffer_load"))], 64, 48, 60)):
        gaps[g] += lst
    if "nodma" in ABL:
        gaps[47][-2] = op("x", "s_nop 0")
    if "nobar" in ABL:
        gaps[47][-1] = op("x", "s_nop 0")
    # statistics of tile t+1
    if "nostat" not in ABL:
        fix = stat_fixup()
        gaps[48] += fix[:2]
        gaps[50] += fix[2:]
        gaps[52] += stat_reads(0)[:4]
        gaps[53] += stat_reads(0)[4:]
        gaps[56] += stat_reads(1)[:4]
        gaps[57] += stat_reads(1)[4:]
    return gaps


def generate(pre):
    e = Emit("bk%d" % (0 if pre else 1))
    # ---------------- prologue
    for kk in range(8):                                              # K / V fragments of this lane's key (rows past Lk: out of range -> 0)
        e(f"buffer_load_dwordx4 {vr(S(0) + 4 * kk, 4)}, %[vokv], %[rk], 0 offen offset:{kk * 32}")
    for kk in range(8):
        e(f"buffer_load_dwordx4 {vr(S(1) + 4 * kk, 4)}, %[vokv], %[rv], 0 offen offset:{kk * 32}")
    e(f"s_mov_b32 {C_STEP}, {SLOT}")
    e(f"s_mov_b32 {C_WRAP}, {(-(NSLOT - 1) * SLOT) & 0xffffffff:#x}")
    e(f"s_mov_b32 {C_STSTEP}, 512")
    e(f"s_mov_b32 {C_STWRAP}, {(-(NSLOT - 1) * 512) & 0xffffffff:#x}")
    for s_ in (S_ROW, S_TR, S_ST):
        e(f"s_mov_b32 {s_}, 0")
    e(f"s_mov_b32 {S_CNT}, %[ntiles]")
    for g in range(3):                                               # tiles 0, 1, 2 -> slots 0, 1, 2
        for ln in dma_group(g * SLOT):
            e(ln)
    e(f"s_mov_b32 {S_DMA}, {2 * SLOT}")
    for kk in range(8):                                              # addresses (attention_bwd2.hip frag_addr)
        e(f"v_xor_b32 {vr(T0)}, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(ROW[kk])}, {vr(T0)}, 5, %[kab]")
    for db in range(4):
        e(f"v_xor_b32 {vr(T0)}, {db}, %[th]")
        e(f"v_lshl_add_u32 {vr(TRA[db])}, {vr(T0)}, 6, %[tab]")
        e(f"v_add_u32 {vr(TRB[db])}, 1024, {vr(TRA[db])}")
        e(f"v_xor_b32 {vr(TRB[db])}, 16, {vr(TRB[db])}")
    e(f"v_mov_b32 {vr(STR)}, %[vstr]")
    e(f"v_mov_b32 {vr(RAW)}, %[vraw]")
    e(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {ar(i)}, 0")
    e("s_waitcnt vmcnt(30)")                                        # the 16 fragment loads (issued first) have landed
    for i in range(32):
        e(f"v_accvgpr_write_b32 {ar(KF(0) + i)}, {vr(S(0) + i)}")
    for i in range(32):
        e(f"v_accvgpr_write_b32 {ar(VF(0) + i)}, {vr(S(1) + i)}")
    e("s_waitcnt vmcnt(20)")                                        # tile 0
    e("s_barrier")
    gaps = tile_ops(pre)
    prep = [o for g in range(48, 64) for o in gaps[g] if o[4]]
    full = linearize(e, prep, [])
    ops = []
    for i in range(64):
        text, needs = mfma_of(i)
        ops.append(op("m", text, needs))
        ops += gaps[i]
    # what is still outstanding where the loop closes (the last MFMAs' waits retired everything older): enter the same way
    loop_pending = linearize(Emit("dry"), ops, full)
    assert full[len(full) - len(loop_pending):] == loop_pending, (full, loop_pending)
    e(f"s_waitcnt lgkmcnt({min(len(loop_pending), 15)})")
    # ---------------- the loop: one tile per iteration
    LOOP = e.lab("loop")
    e.label(LOOP)
    end = linearize(e, ops, loop_pending)
    assert end == loop_pending, (end, loop_pending)
    e(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    e(f"s_cmp_lg_u32 {S_CNT}, 0")
    e(f"s_cbranch_scc1 {LOOP}")
    # ---------------- epilogue: accumulators -> this wave's part of the (now idle) ring; the HIP code takes them from there
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    e("s_nop 7")
    e("s_nop 7")
    for x in range(8):
        for g in range(4):
            e(f"ds_write_b128 %[vdump], {ar(x * 16 + 4 * g, 4)} offset:{(x * 4 + g) * PARK_BLOCK}")
    e("s_waitcnt lgkmcnt(0)")
    return e



# ================================================================================================ the dQ stream
# attn_bwd2_dq_w64_kernel: 4 waves x 32 queries (lane = query), loop over tiles of 64 keys whose K and V rows are LDS-DMA
# staged like Q / dO above.  Per 32-key half:  A  S^T = K Q^T, dP^T = V dO^T (16 MFMA; the accumulators start from 16
# copies of -lse' / -delta: MFMA C operand, no statistics traffic);  B  dS = exp2(S) dP (P itself is not needed);
# C  dQ^T += K^T dS^T (8 MFMA, transposed fragments of the K tile).  48 MFMAs per tile:  A0 | A1 + B0 | C0 + B1 | C1 + B1.
# The keys past klen of the LAST tile get S = -inf (a block of 32 v_cndmask per half, skipped by every other tile).
#   a[0:63] dQ^T accumulators   a[64:95] Q fragments   a[96:127] dO fragments
#   v[160:175] -lse' x 16   v[176:191] -delta x 16   v[128:143] packed dS [half][a]   v[192:255] fragment ring, 16 slots of 4
def DQA(db):       return db * 16
def QF(kk):        return 64 + kk * 4
def DOF(kk):       return 96 + kk * 4
NEGL, NEGD = 160, 176
def PKD(hb, a):    return 128 + hb * 8 + a * 4
def SLOT16(i):     return 192 + (i % 16) * 4
VLIM = 40
S_REM = "%[srem]"
DQ_LEAD_A, DQ_LEAD_C = 12, 6


def dq_mfma_of(i):
    frag = vr(SLOT16(i), 4)
    if i < 32:
        hb, kk, w = i // 16, (i % 16) // 2, i % 2
        acc = vr((S(hb) if w == 0 else DP(hb)), 16)
        c = vr((NEGL if w == 0 else NEGD), 16) if kk == 0 else acc
        b = ar((QF(kk) if w == 0 else DOF(kk)), 4)
        return f"v_mfma_f32_32x32x16_bf16 {acc}, {frag}, {b}, {c}", ([] if "norow" in ABL else [f"F{i}"])
    j = i - 32
    hb, a, db = j // 8, (j % 8) // 4, j % 4
    acc = ar(DQA(db), 16)
    return f"v_mfma_f32_32x32x16_bf16 {acc}, {frag}, {vr(PKD(hb, a), 4)}, {acc}", ([] if "notr" in ABL else [f"F{i}a", f"F{i}b"])


def dq_frag_reads(i, prep):
    if ("norow" in ABL and i < 32) or ("notr" in ABL and i >= 32):
        return []
    if i < 32:
        hb, kk, w = i // 16, (i % 16) // 2, i % 2
        off = hb * 8192 + (16384 if w == 1 else 0)              # S: K rows;  dP: V rows
        return [op("r", f"ds_read_b128 {vr(SLOT16(i), 4)}, {vr(ROW[kk])} offset:{off}", tag=f"F{i}", prep=prep)]
    j = i - 32
    hb, a, db = j // 8, (j % 8) // 4, j % 4
    off = hb * 8192 + a * 4096                                  # K^T of keys 32 hb + 16 a ..
    return [op("r", f"ds_read_b64_tr_b16 {vr(SLOT16(i), 2)}, {vr(TRA[db])} offset:{off}", tag=f"F{i}a", prep=prep),
            op("r", f"ds_read_b64_tr_b16 {vr(SLOT16(i) + 2, 2)}, {vr(TRB[db])} offset:{off}", tag=f"F{i}b", prep=prep)]


def dq_stage_b(hb, pre):
    out = []
    for ee in range(8):
        r0, r1 = 2 * ee, 2 * ee + 1
        if not pre:
            out += [f"v_mul_f32 {vr(S(hb) + r0)}, %[sc], {vr(S(hb) + r0)}", f"v_mul_f32 {vr(S(hb) + r1)}, %[sc], {vr(S(hb) + r1)}"]
        out += [f"v_exp_f32 {vr(S(hb) + r0)}, {vr(S(hb) + r0)}",
                f"v_exp_f32 {vr(S(hb) + r1)}, {vr(S(hb) + r1)}",
                f"v_mul_f32 {vr(DP(hb) + r0)}, {vr(S(hb) + r0)}, {vr(DP(hb) + r0)}",
                f"v_mul_f32 {vr(DP(hb) + r1)}, {vr(S(hb) + r1)}, {vr(DP(hb) + r1)}",
                f"v_cvt_pk_bf16_f32 {vr(PKD(hb, ee // 4) + ee % 4)}, {vr(DP(hb) + r0)}, {vr(DP(hb) + r1)}"]
    return [op("x", ln) for ln in out]


def dq_mask_block(e, hb, tag):
    """Last tile only: keys at or past klen of half hb get S = -inf (register r <-> key 32 hb + 16 (r>>3) + 8 lh + (r&7))."""
    skip = e.lab(f"nomask_{tag}")
    out = [f"s_cmp_ge_i32 {S_REM}, 64", f"s_cbranch_scc1 {skip}", "s_nop 7", "s_nop 7",
           f"v_sub_u32 {vr(VLIM)}, {S_REM}, %[lh8]"]              # keys left, seen from this lane's 8 lh offset
    for r in range(16):
        base = 32 * hb + 16 * (r >> 3) + (r & 7)
        out += [f"v_cmp_lt_i32 vcc, {base}, {vr(VLIM)}",            # key in range: keep; else -inf
                f"v_cndmask_b32 {vr(S(hb) + r)}, {vr(NEGINF)}, {vr(S(hb) + r)}, vcc"]
    out.append(f"{skip}:")
    return out


def dq_dma_group(slot_expr):
    out = []
    for which, rs in enumerate(("%[rk]", "%[rv]")):
        for j in range(4):
            if j == 0:
                out.append(f"s_add_u32 m0, %[ldsw], {slot_expr}" if which == 0 else "s_add_u32 m0, m0, 0x1000")
                out.append(f"s_mov_b32 {S_OFF}, %[skn]")
            else:
                out.append("s_add_u32 m0, m0, 0x1000")
                out.append(f"s_add_u32 {S_OFF}, {S_OFF}, %[skp]")
            out.append(f"buffer_load_dwordx4 %[vodk], {rs}, {S_OFF} offen lds")
    out.append(f"s_add_u32 %[skn], {S_OFF}, %[skp]")                # next tile
    return out


def dq_tile_ops(pre, e):
    gaps = [[] for _ in range(48)]
    for tgt in range(48, 96):
        lead = DQ_LEAD_A if tgt % 48 < 32 else DQ_LEAD_C
        g = tgt - lead
        gaps[g % 48] += dq_frag_reads(tgt % 48, prep=g < 48)
        assert (tgt % 48) - 16 <= (tgt % 48) - lead
    b0, b1 = dq_stage_b(0, pre), dq_stage_b(1, pre)
    if "nob" in ABL:
        b0, b1 = [], []
    h0, h1 = len(b0) // 2, len(b1) // 2
    for lst, lo, hi in ((b0[:h0], 17, 28), (b0[h0:], 28, 35), (b1[:h1], 33, 39), (b1[h1:], 39, 43)):
        for g, ops_ in enumerate(spread(lst, 48, lo, hi)):
            gaps[g] += ops_
    # the edge masks sit right in front of each half's stage B (multi-line ops keep the labels with their branches)
    if "nomask" not in ABL:
        gaps[16] += [op("x", "\\n\\t\"\n    \"".join(dq_mask_block(e, 0, "h0")))]
        gaps[32] += [op("x", "\\n\\t\"\n    \"".join(dq_mask_block(e, 1, "h1")))]
    gaps[20] += [op("x", ln) for ln in advance(S_ROW, D_ROW, C_STEP, C_WRAP, NSLOT * SLOT)]
    for g, lst in enumerate(spread([op("x", f"v_add_u32 {vr(r)}, {D_ROW}, {vr(r)}") for r in ROW], 48, 21, 33)):
        gaps[g] += lst
    gaps[43] += [op("x", ln) for ln in advance(S_TR, D_TR, C_STEP, C_WRAP, NSLOT * SLOT)]
    for g, lst in enumerate(spread([op("x", f"v_add_u32 {vr(r)}, {D_TR}, {vr(r)}") for r in TRA + TRB], 48, 44, 48)):
        gaps[g] += lst
    # barrier where the next tile's first fragments are requested (gap 36 = 48 - 12); the 8 loads of tile t+2 may still fly
    gaps[35] += [op("x", "s_waitcnt vmcnt(8)"), op("x", "s_barrier"), op("x", f"s_sub_u32 {S_REM}, {S_REM}, 64")]
    gaps[36] += [op("x", ln) for ln in advance(S_DMA, S_OFF, C_STEP, C_WRAP, NSLOT * SLOT)[:2] + [f"s_cselect_b32 {S_DMA}, 0, {S_DMA}"]]
    for g, lst in enumerate(spread([op("x", ln) for ln in dq_dma_group(S_DMA) if not ("nodma" in ABL and ln.startswith("buffer_load"))], 48, 36, 46)):
        gaps[g] += lst
    if "nodma" in ABL:
        gaps[35][0] = op("x", "s_nop 0")
    return gaps


def generate_dq(pre):
    e = Emit("bq%d" % (0 if pre else 1))
    for kk in range(8):                                              # Q / dO fragments of this lane's query (rows past Lq: out of range -> 0)
        e(f"buffer_load_dwordx4 {vr(S(0) + 4 * kk, 4)}, %[voq], %[rq], 0 offen offset:{kk * 32}")
    for kk in range(8):
        e(f"buffer_load_dwordx4 {vr(S(1) + 4 * kk, 4)}, %[vodof], %[rdo], 0 offen offset:{kk * 32}")
    e(f"s_mov_b32 {C_STEP}, {SLOT}")
    e(f"s_mov_b32 {C_WRAP}, {(-(NSLOT - 1) * SLOT) & 0xffffffff:#x}")
    for s_ in (S_ROW, S_TR):
        e(f"s_mov_b32 {s_}, 0")
    e(f"s_mov_b32 {S_CNT}, %[ntiles]")
    for g in range(3):
        for ln in dq_dma_group(g * SLOT):
            e(ln)
    e(f"s_mov_b32 {S_DMA}, {2 * SLOT}")
    for kk in range(8):
        e(f"v_xor_b32 {vr(T0)}, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(ROW[kk])}, {vr(T0)}, 5, %[kab]")
    for db in range(4):
        e(f"v_xor_b32 {vr(T0)}, {db}, %[th]")
        e(f"v_lshl_add_u32 {vr(TRA[db])}, {vr(T0)}, 6, %[tab]")
        e(f"v_add_u32 {vr(TRB[db])}, 1024, {vr(TRA[db])}")
        e(f"v_xor_b32 {vr(TRB[db])}, 16, {vr(TRB[db])}")
    e(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
    for r in range(16):
        e(f"v_mov_b32 {vr(NEGL + r)}, %[negl]")
        e(f"v_mov_b32 {vr(NEGD + r)}, %[negd]")
    for i in range(64):
        e(f"v_accvgpr_write_b32 {ar(i)}, 0")
    e("s_waitcnt vmcnt(24)")                                        # the 16 fragment loads (issued first) have landed
    for i in range(32):
        e(f"v_accvgpr_write_b32 {ar(QF(0) + i)}, {vr(S(0) + i)}")
    for i in range(32):
        e(f"v_accvgpr_write_b32 {ar(DOF(0) + i)}, {vr(S(1) + i)}")
    e("s_waitcnt vmcnt(16)")                                        # tile 0
    e("s_barrier")
    gaps = dq_tile_ops(pre, e)
    prep = [o for g in range(36, 48) for o in gaps[g] if o[4]]
    full = linearize(e, prep, [])
    ops = []
    for i in range(48):
        text, needs = dq_mfma_of(i)
        ops.append(op("m", text, needs))
        ops += gaps[i]
    loop_pending = linearize(Emit("dry"), ops, full)
    assert full[len(full) - len(loop_pending):] == loop_pending, (full, loop_pending)
    e(f"s_waitcnt lgkmcnt({min(len(loop_pending), 15)})")
    LOOP = e.lab("loop")
    e.label(LOOP)
    end = linearize(e, ops, loop_pending)
    assert end == loop_pending, (end, loop_pending)
    e(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    e(f"s_cmp_lg_u32 {S_CNT}, 0")
    e(f"s_cbranch_scc1 {LOOP}")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    e("s_nop 7")
    e("s_nop 7")
    for x in range(4):
        for g in range(4):
            e(f"ds_write_b128 %[vdump], {ar(x * 16 + 4 * g, 4)} offset:{(x * 4 + g) * PARK_BLOCK}")
    e("s_waitcnt lgkmcnt(0)")
    return e


CLOBBER_V = range(16, 256)
CLOBBER_A = range(0, 256)
CLOBBER_S = range(80, 93)


def main():
    print("// GENERATED by gen_attn_bwd_w64.py — do not edit; edit the generator.")
    for name, pre in (("PRE", True), ("GEN", False)):
        e = generate(pre)
        print(f"#define OMH_ATTN_BWD_W64_ASM_{name} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        n_mfma = sum("v_mfma" in ln for ln in e.lines)
        print(f"// {name}: {len(e.lines)} lines, {n_mfma} MFMA in the loop body")
    for name, pre in (("PRE", True), ("GEN", False)):
        e = generate_dq(pre)
        print(f"#define OMH_ATTN_BWD_DQ_W64_ASM_{name} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        print(f"// dQ {name}: {len(e.lines)} lines")
    print(f"#define OMH_ATTN_BWD_DQ_W64_LDS {NSLOT * SLOT}")
    print(f"#define OMH_ATTN_BWD_W64_LDS {LDS_BYTES}")
    print(f"#define OMH_ATTN_BWD_W64_SLOT {SLOT}")
    print(f"#define OMH_ATTN_BWD_W64_STAT_RAW {STAT_RAW}")
    print(f"#define OMH_ATTN_BWD_W64_STAT_FIN {STAT_FIN}")
    print(f"#define OMH_ATTN_BWD_W64_PARK_BLOCK {PARK_BLOCK}")
    print(f"#define OMH_ATTN_BWD_W64_PARK_WAVE {PARK_WAVE}")
    assert 4 * PARK_WAVE <= LDS_BYTES
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in CLOBBER_S] + [f'"v{i}"' for i in CLOBBER_V] + \
           [f'"a{i}"' for i in CLOBBER_A]
    print("#define OMH_ATTN_BWD_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))


if __name__ == "__main__":
    main()
