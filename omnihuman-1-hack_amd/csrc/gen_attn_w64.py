#!/usr/bin/env python3
"""Generator of the instruction streams of ``flash_attn_fwd_d128_w64_kernel<V>`` (attention_w64.hip).

The long-sequence attention kernel of round 2: 4 waves x 64 query rows per workgroup, ONE wave per SIMD owning
the whole 512-entry register file.  hipcc cannot be made to keep 128 accumulators + 64 operand registers in the
accumulator half without shuffling them through v_accvgpr_read/write (round 1, DESIGN.md 4.4 last row: 744 TF), so
the kernel body is written out instruction by instruction here, with a fixed register map, and pasted into the
kernel as one ``asm volatile`` block (``attention_w64_asm.inc``).  The surrounding HIP code computes addresses
and descriptors only.

    python gen_attn_w64.py > attention_w64_asm.inc        (build.py does this when the generator changed)

Math and layouts are those of attention.hip (S^T = K Q^T, O^T = V^T P^T, key bits 2<->3 swapped so P never moves
between lanes; K [64][128] / V^T [128][64] bf16 tiles, 16-byte-slot XOR swizzle, LDS-DMA staged).  What is new:

  * two 32-query blocks per wave share every K / V^T fragment read: 0.5 ds_read_b128 per MFMA instead of 1;
  * O (128 regs) and the pre-scaled Q fragments (64 regs) live in AGPRs, K / V^T fragment rings too — the arch
    VGPRs hold two score sets (128), the packed P (32) and the softmax state;
  * q carries scale*log2(e) (folded into the norm kernel, or applied in the prologue here), and the running max
    enters the scores through the MFMA's C operand (16 registers per query block holding -m): p = exp2(S) needs NO
    per-element subtract / fma;
  * the running max is only moved when a row's new maximum exceeds it by more than THR = 4 (factor 16 in P):
    the rescale of O (AGPR round trip) is a rare slow path, never on the hot path;
  * per tile: phase 1 = K(t+1).Q^T (32 MFMA) || exp2 + bf16 packing of tile t (96 VALU) || K fragment reads,
    phase 2 = V^T(t).P^T(t) (32 MFMA) || row sums of tile t + row max of tile t+1 (~100 VALU) || V^T fragment reads;
    <= 4.2 non-MFMA issues per MFMA (the budget is 5: MI355X_MICROARCH.md, one wave per SIMD).

Variants (selected at launch, OMH_W64_VARIANT; kept side by side for A/B timing on one box):
  V0  K / V^T rings of two 16 KiB slots each, all LDS-DMA issued at the head of phase 1, every phase starts by
      reading its own first fragments;
  V1  = V0 with the eight DMA pieces spread over phase 1 and the first three V^T fragments of phase 2 read under
      the last MFMAs of phase 1;
  V2  = V1 with a K ring of THREE slots filled three tiles ahead, so that the first K fragments of the next tile are
      read under the last MFMAs of phase 2 (no phase starts by waiting for the LDS);
  V3  = V2 storing its normalised result in fp32: the split-KV workers of the tail (attention_w64.hip).

Register map (asm-owned; the thirteen per-lane inputs stay in the compiler's operand registers v0..v11):
  a[0:127]    O^T accumulators   [qb][db] 16 each
  a[128:191]  Q fragments        [qb][kk] 4 each (bf16x8)
  a[192:215]  K fragment ring    3 stages x [kb] x 4
  a[216:231]  V^T fragment ring  4 stages x 4
  v[12:19]    K fragment LDS addresses (per kk)      v[20:23]  V^T fragment LDS addresses (per kb, a)
  v[32:95]    score set 0        [qb][kb] 16 each    v[96:159] score set 1
  v[160:191]  MI = -m_run        [qb] 16 copies (MFMA C operand)
  v[192:223]  P packed bf16      [qb][kb][a] 4 each
  v[224:255]  state / temporaries
  s92 s93 DMA source offsets, s94 tile counter, s95 K tile stride, s[96:97] exec save, s98 K read-slot offset,
  s99 K DMA-slot offset, s91 address step of the K ring
"""
import sys

THR = "4.0"                      # inline constant: rescale when a row max exceeds the running max by > 4 (log2 units)
KSLOT = 16384


# ---------------------------------------------------------------- register map
def O(qb, db):      return (qb * 4 + db) * 16
def Q(qb, kk):      return 128 + (qb * 8 + kk) * 4
def KR(stage, kb):  return 192 + stage * 8 + kb * 4
def VR(stage):      return 216 + stage * 4
def S(st, qb, kb):  return 32 + st * 64 + (qb * 2 + kb) * 16
def MI(qb):         return 160 + qb * 16
def P(qb, kb, a):   return 192 + ((qb * 2 + kb) * 2 + a) * 4


KA = list(range(12, 20))
VA = list(range(20, 24))
M_RUN = [224, 225]
L_A = [226, 227]
L_B = [228, 229]
MXA = [230, 231]
MXB = [232, 233]
T0, T1, T2, T3 = 234, 235, 236, 237
NEGINF = 238
VLIM = 239
INV = [240, 241]
E0 = 242                          # 242..249 scratch (8)


def vr(lo, n=1):   return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"
def ar(lo, n=1):   return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


class Emit:
    def __init__(self, tag):
        self.lines, self.tag = [], tag

    def __call__(self, s):
        self.lines.append(s)

    def lab(self, name):
        return f".Lw64v{self.tag}_{name}_%="

    def label(self, name):
        self.lines.append(f"{name}:")

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


# ---------------------------------------------------------------- op lists and the LDS wait tracker
# op = ("m", text, [tags needed])  MFMA   |  ("r", tag, text)  ds_read   |  ("x", text)  anything else
def linearize(e, ops, pending):
    """Emit `ops` in order; before an MFMA that needs ds_read results, emit the s_waitcnt lgkmcnt that makes exactly
    those reads complete (LDS returns in order).  `pending` = tags of reads issued earlier and not yet waited for,
    oldest first.  Returns the pending list at the end."""
    pending = list(pending)
    for op in ops:
        if op[0] == "r":
            e(op[2])
            pending.append(op[1])
        elif op[0] == "m":
            need = [t for t in op[2] if t in pending]
            if need:
                last = max(pending.index(t) for t in need)
                allowed = len(pending) - last - 1
                assert allowed <= 15
                e(f"s_waitcnt lgkmcnt({allowed})")
                pending = pending[last + 1:]
            for ln in (op[1] if isinstance(op[1], tuple) else (op[1],)):      # (M16 ablation: two instructions per op)
                e(ln)
        else:
            e(op[1])
    return pending


def weave(mfmas, plans, pre=()):
    """mfmas: list of ("m", ...) ops; plans: lists (one entry per gap) of lists of ops issued after that MFMA."""
    ops = list(pre)
    for g, m in enumerate(mfmas):
        ops.append(m)
        for pl in plans:
            ops.extend(pl[g])
    return ops


def spread(items, gaps, start=0, end=None):
    """Distribute `items` (ordered) over gaps[start:end] as evenly as possible; returns a per-gap list of lists."""
    end = gaps if end is None else end
    n = end - start
    out = [[] for _ in range(gaps)]
    for i, it in enumerate(items):
        out[start + min(n - 1, (i * n) // len(items))].append(it)
    return out


def X(lines):
    return [("x", ln) for ln in lines]


# ---------------------------------------------------------------- pieces
def kread_ops(kk, kbase):
    st = kk % 3
    return [("r", f"K{kk}.0", f"ds_read_b128 {ar(KR(st, 0), 4)}, {vr(KA[kk])} offset:{kbase}"),
            ("r", f"K{kk}.1", f"ds_read_b128 {ar(KR(st, 1), 4)}, {vr(KA[kk])} offset:{kbase + 8192}")]


def vread_op(i, vbase):
    kb, a, db = i >> 3, (i >> 2) & 1, i & 3
    return ("r", f"V{i}", f"ds_read_b128 {ar(VR(i % 4), 4)}, {vr(VA[2 * kb + a])} offset:{vbase + db * 4096}")


M16 = False     # OMH_ATTN_ABL=4, variant 0 (timing only, round 6): every 32x32x16 MFMA as TWO v_mfma_f32_16x16x32_bf16 on the
#                 same operand registers — the same flops and traffic, garbage results: what would the MFMA shape that sustains
#                 2.16 instead of 1.88 PFLOP/s on data-like operands (tools/mfma_shape_probe.py) buy this stream?


def qk_mfmas(nxt, c_zero=False):
    """32 MFMAs of K.Q^T into score set nxt (C = MI = -m_run on the first k step), + the fragment reads that ride
    behind them (steps kk+2, after the 2nd MFMA of step kk)."""
    mf = []
    for kk in range(8):
        for kb, qb in ((0, 0), (0, 1), (1, 0), (1, 1)):
            if M16:
                two = []
                for half in range(2):
                    d = vr(S(nxt, qb, kb) + 4 * half, 4)
                    c = ("0" if c_zero else vr(MI(qb) + 4 * half, 4)) if kk == 0 else d
                    two.append(f"v_mfma_f32_16x16x32_bf16 {d}, {ar(KR(kk % 3, kb), 4)}, {ar(Q(qb, kk), 4)}, {c}")
                mf.append(("m", tuple(two), [f"K{kk}.{kb}"]))
                continue
            c = ("0" if c_zero else vr(MI(qb), 16)) if kk == 0 else vr(S(nxt, qb, kb), 16)
            mf.append(("m", f"v_mfma_f32_32x32x16_bf16 {vr(S(nxt, qb, kb), 16)}, {ar(KR(kk % 3, kb), 4)}, "
                            f"{ar(Q(qb, kk), 4)}, {c}", [f"K{kk}.{kb}"]))
    return mf


def qk_read_plan(kbase, first=2):
    plan = [[] for _ in range(32)]
    for kk in range(8):
        if first <= kk + 2 < 8:
            plan[kk * 4 + 1] = kread_ops(kk + 2, kbase)
    return plan


# Timing-only ablation builds (tools/attn_abl.sh; wrong numerics by construction, never shipped — the default output of
# this generator is unchanged).  OMH_ATTN_ABL=1: variant 0 = V2 with every 4th exponential replaced by the 5 FMA-pipe
# instructions a degree-3 exp2 polynomial costs (floor, subtract, 3 fma; the exponent insertion not even counted),
# variant 1 = V2 without the per-piece SALU address arithmetic of the LDS-DMA (all four pieces of a tile then fetch the
# same rows into the same 1 KiB).  OMH_ATTN_ABL=2: variant 1 = V2 with that arithmetic kept but adding zero (same
# instruction count, same-rows traffic).  Round-3 results (profiles/r03_attention_ablations.json): polynomial share
# -3.5 % (slower); no-SALU +7.7 %, same-count same-rows +10.3 % — i.e. NOT the SALU slots: with three quarters of every
# K / V^T tile left stale the operands toggle less and the power-limited chip clocks higher (see the telemetry there).
ABL = __import__("os").environ.get("OMH_ATTN_ABL", "")


def softmax_lines(cur, poly=False):
    """exp2 in place + bf16 packing of score set cur: 64 + 32 VALU; a block's packing trails its exponentials."""
    out = []
    for qb in range(2):
        for kb in range(2):
            base = S(cur, qb, kb)
            for r in range(16):
                if poly and r % 4 == 3:
                    x = vr(base + r)
                    out += [f"v_floor_f32 {vr(T3)}, {x}", f"v_sub_f32 {x}, {x}, {vr(T3)}", f"v_fma_f32 {vr(T3)}, {x}, 0.5, 1.0",
                            f"v_fma_f32 {vr(T3)}, {x}, {vr(T3)}, 0.5", f"v_fma_f32 {x}, {x}, {vr(T3)}, 1.0"]
                else:
                    out.append(f"v_exp_f32 {vr(base + r)}, {vr(base + r)}")
            for a in range(2):
                for eidx in range(4):
                    r0 = base + 8 * a + 2 * eidx
                    out.append(f"v_cvt_pk_bf16_f32 {vr(P(qb, kb, a) + eidx)}, {vr(r0)}, {vr(r0 + 1)}")
    return out


def rowsum_lines(cur):
    # (round 5: the same sums as 32 v_pk_add_f32 on (L_A, L_B) register pairs — bit-identical, 32 VALU instructions fewer per
    #  tile — measured SLOWER, 4.88 -> 5.07 ms per launch on one box: packed fp32 adds do not issue faster here and the two
    #  accumulator chains end up two instructions apart instead of four).  Likewise 32 v_dot2_f32_bf16 on the PACKED bf16 pairs
    #  (P.lo * 1 + P.hi * 1 + acc): 4.82 -> 4.99 ms.  VOP3P instructions cost ~3x a plain VALU instruction in this stream.)
    a, b = [], []
    for qb, dst in ((0, a), (1, b)):
        regs = [S(cur, qb, kb) + r for kb in range(2) for r in range(16)]
        for i, r in enumerate(regs):
            acc = L_A[qb] if i % 2 == 0 else L_B[qb]
            dst.append(f"v_add_f32 {vr(acc)}, {vr(acc)}, {vr(r)}")
    mixed = []
    for i in range(0, 32, 2):                                   # dependent adds end up 4 apart
        mixed += [a[i], a[i + 1], b[i], b[i + 1]]
    return mixed


def rowmax_lines(nxt):
    """Row max of score set nxt per query block -> MXA[qb] (both half-waves hold the row's max)."""
    out = []
    for qb in range(2):
        s0, s1 = S(nxt, qb, 0), S(nxt, qb, 1)
        out.append(f"v_max_f32 {vr(MXA[qb])}, {vr(s0)}, {vr(s1)}")
        out.append(f"v_max_f32 {vr(MXB[qb])}, {vr(s0 + 1)}, {vr(s1 + 1)}")
    for r in range(2, 16, 2):
        for qb in range(2):
            s0, s1 = S(nxt, qb, 0), S(nxt, qb, 1)
            out.append(f"v_max3_f32 {vr(MXA[qb])}, {vr(MXA[qb])}, {vr(s0 + r)}, {vr(s1 + r)}")
            out.append(f"v_max3_f32 {vr(MXB[qb])}, {vr(MXB[qb])}, {vr(s0 + r + 1)}, {vr(s1 + r + 1)}")
    for qb in range(2):
        out.append(f"v_max_f32 {vr(MXA[qb])}, {vr(MXA[qb])}, {vr(MXB[qb])}")
    for qb in range(2):
        out.append(f"v_mov_b32 {vr(MXB[qb])}, {vr(MXA[qb])}")
    out.append("s_nop 1")
    for qb in range(2):
        out.append(f"v_permlane32_swap_b32 {vr(MXA[qb])}, {vr(MXB[qb])}")
    for qb in range(2):
        out.append(f"v_max_f32 {vr(MXA[qb])}, {vr(MXA[qb])}, {vr(MXB[qb])}")
    return out


def pv_mfmas(cur):
    mf = []
    for i in range(16):
        kb, a, db = i >> 3, (i >> 2) & 1, i & 3
        for qb in range(2):
            if M16:
                two = []
                for half in range(2):
                    d = ar(O(qb, db) + 4 * half, 4)
                    two.append(f"v_mfma_f32_16x16x32_bf16 {d}, {ar(VR(i % 4), 4)}, {vr(P(qb, kb, a), 4)}, {d}")
                mf.append(("m", tuple(two), [f"V{i}"]))
                continue
            mf.append(("m", f"v_mfma_f32_32x32x16_bf16 {ar(O(qb, db), 16)}, {ar(VR(i % 4), 4)}, "
                            f"{vr(P(qb, kb, a), 4)}, {ar(O(qb, db), 16)}", [f"V{i}"]))
    return mf


def pv_read_plan(vbase, first=3):
    plan = [[] for _ in range(32)]
    for i in range(16):
        if first <= i + 3 < 16:
            plan[2 * i] = [vread_op(i + 3, vbase)]
    return plan


def mask_block(e, st):
    """Scores of set `st` whose key index >= klen -> -inf.  %[srem] = klen - 64 * tile (1..63 when the block runs)."""
    e(f"v_lshlrev_b32 {vr(VLIM)}, 3, %[lh]")
    e(f"v_sub_u32 {vr(VLIM)}, %[srem], {vr(VLIM)}")           # rem - 8 h : mask register r when c(r) >= vlim
    for kb in range(2):
        for r in range(16):
            c = 32 * kb + 16 * (r >> 3) + (r & 7)
            e(f"v_cmp_ge_i32 vcc, {c}, {vr(VLIM)}")
            for qb in range(2):
                reg = S(st, qb, kb) + r
                e(f"v_cndmask_b32 {vr(reg)}, {vr(reg)}, {vr(NEGINF)}, vcc")


def slow_path(e, st, qb, first):
    """Move the running max of query block qb: delta = first ? rowmax : max(rowmax, 0) (rowmax is relative to the
    running max already), then O, l, the pending scores of set `st` and -m (MI) follow."""
    d, al = T0, T1
    if first:
        e(f"v_mov_b32 {vr(d)}, {vr(MXA[qb])}")
    else:
        e(f"v_max_f32 {vr(d)}, 0, {vr(MXA[qb])}")
    e(f"v_add_f32 {vr(M_RUN[qb])}, {vr(M_RUN[qb])}, {vr(d)}")
    e(f"v_sub_f32 {vr(al)}, 0, {vr(d)}")
    e(f"v_exp_f32 {vr(al)}, {vr(al)}")
    for kb in range(2):
        for r in range(16):
            reg = S(st, qb, kb) + r
            e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr(d)}")
    for r in range(16):
        e(f"v_sub_f32 {vr(MI(qb) + r)}, {vr(MI(qb) + r)}, {vr(d)}")
    if not first:
        e(f"v_mul_f32 {vr(L_A[qb])}, {vr(L_A[qb])}, {vr(al)}")
        e(f"v_mul_f32 {vr(L_B[qb])}, {vr(L_B[qb])}, {vr(al)}")
        e("s_nop 7")
        e("s_nop 7")
        e("s_nop 7")                                           # MFMA write of O -> v_accvgpr_read
        for db in range(4):
            for r in range(0, 16, 4):
                regs = [O(qb, db) + r + i for i in range(4)]
                tmp = [E0 + i for i in range(4)]
                for a_, t_ in zip(regs, tmp):
                    e(f"v_accvgpr_read_b32 {vr(t_)}, {ar(a_)}")
                for t_ in tmp:
                    e(f"v_mul_f32 {vr(t_)}, {vr(t_)}, {vr(al)}")
                for a_, t_ in zip(regs, tmp):
                    e(f"v_accvgpr_write_b32 {ar(a_)}, {vr(t_)}")
    e("s_nop 7")                                               # VALU write -> MFMA read (MI as C, O as C)


def check_and_rescale(e, st, tag, first=False):
    """Per query block: move the running max when the new row max exceeds it by more than THR.  ONE branch skips both
    blocks' checks in the common case (round 5: a taken branch costs a one-wave-per-SIMD stream ~40 cycles of instruction
    fetch, and there were two per tile); vccz is stale after a scalar write of vcc, so the merged test goes through scc."""
    if first:
        for qb in range(2):
            slow_path(e, st, qb, True)
        return
    both = e.lab(f"nores_{tag}")
    e(f"v_cmp_lt_f32 vcc, {THR}, {vr(MXA[0])}")
    e("s_mov_b64 s[96:97], vcc")
    e(f"v_cmp_lt_f32 vcc, {THR}, {vr(MXA[1])}")
    e("s_or_b64 s[96:97], s[96:97], vcc")
    e("s_cmp_eq_u64 s[96:97], 0")
    e(f"s_cbranch_scc1 {both}")
    for qb in range(2):
        skip = e.lab(f"nores_{tag}_{qb}")
        e(f"v_cmp_lt_f32 vcc, {THR}, {vr(MXA[qb])}")
        e(f"s_cbranch_vccz {skip}")
        slow_path(e, st, qb, False)
        e.label(skip)
    e.label(both)


def rotate(sreg, nslots):
    return [f"s_add_u32 {sreg}, {sreg}, {KSLOT}", f"s_cmp_eq_u32 {sreg}, {nslots * KSLOT}", f"s_cselect_b32 {sreg}, 0, {sreg}"]


# ---------------------------------------------------------------- whole stream
def generate(variant):
    abl_poly = ABL == "1" and variant == 0
    abl_salu = ABL == "1" and variant == 1
    abl_traffic = ABL == "2" and variant == 1
    # OMH_ATTN_ABL=3 (round 5): variant 0 = V2 without the row maxima and the rescale check, variant 1 = without the row sums
    # as well.  Measured: 4.707 -> 4.555 -> 4.411 ms per launch (-3.2 %, -6.3 %).  A stream without the running max would
    # need an a-priori bound on the scores; the model's RMS norm runs over all 1 536 channels, not per head, so the
    # worst-case bound is 196 (log2 units) instead of 16 — not usable (p would underflow), not built.
    abl_nomax = ABL == "3" and variant in (0, 1)
    abl_nosum = ABL == "3" and variant == 1
    global M16
    M16 = ABL == "4" and variant == 0
    base = 2 if (abl_poly or abl_salu or abl_traffic or abl_nomax or M16) else min(variant, 2)
    dma_spread = base >= 1
    vpre = base >= 1
    k3 = base >= 2
    # (timing-only ablations of V2 measured in round 2, profiles/r02_attention_w64_ablations.json: no LDS-DMA in the
    # loop -5.9 %, v_exp -> v_mov -3.4 %, no barrier -0.2 %, no ds_reads -8.1 %: nothing dominates)
    abl = 0
    # variant 3 = variant 2 with the "partial" epilogue of the split-KV tail workers: the normalised output is stored
    # in fp32 (one [256, 128] slab per worker) instead of bf16; with the log-sum-exp stored next to it a combine
    # pass weights the workers' results (attention_w64.hip)
    partial = variant == 3
    # LDS map: two-slot rings: K [0, 32K) | V^T [32K, 64K).  Three-slot K ring: V^T [0, 32K) | K [32K, 80K) — the K
    # addresses carry the ring base and slot in the address registers, so every ds_read offset stays below 64 KiB
    VBASE = 0 if k3 else 2 * KSLOT
    KBASE = 2 * KSLOT if k3 else 0
    e = Emit(variant)

    def k_dma(slot_expr, advance=True):
        """4 pieces of K(next) -> K ring slot; slot_expr: literal byte offset or an SGPR name."""
        out = [f"s_add_u32 m0, %[ldsw], {slot_expr}"]
        if KBASE:
            out.append(f"s_add_u32 m0, m0, {KBASE}")
        for j in range(4):
            if j and not abl_salu:
                out.append("s_add_u32 m0, m0, 0" if abl_traffic else "s_add_u32 m0, m0, 4096")
            if not (abl_salu and j):
                out.append("s_mov_b32 s92, %[skn]" if j == 0 else ("s_add_u32 s92, s92, 0" if abl_traffic else "s_add_u32 s92, s92, %[skp]"))
            out.append("buffer_load_dwordx4 %[vok], %[rk], s92 offen lds")
        if advance:
            out.append("s_add_u32 %[skn], %[skn], s95")
        return out

    def v_dma(slot):
        out = [f"s_add_u32 m0, %[ldsw], {VBASE + slot * KSLOT}"]
        for j in range(4):
            if j and not abl_salu:
                out.append("s_add_u32 m0, m0, 0" if abl_traffic else "s_add_u32 m0, m0, 4096")
            if not (abl_salu and j):
                out.append("s_mov_b32 s93, %[svn]" if j == 0 else ("s_add_u32 s93, s93, 0" if abl_traffic else "s_add_u32 s93, s93, %[svp]"))
            out.append("buffer_load_dwordx4 %[vov], %[rv], s93 offen lds")
        out.append("s_add_u32 %[svn], %[svn], 128")
        return out

    # ---------------- prologue: Q fragments -> (pre-scaled) bf16 -> AGPRs
    for qb in range(2):
        for kk in range(8):
            so = "0" if qb == 0 else "%[sq1]"
            e(f"buffer_load_dwordx4 {vr(32 + (qb * 8 + kk) * 4, 4)}, %[voq], %[rq], {so} offen offset:{kk * 32}")
    e("s_lshl_b32 s95, %[skp], 2")
    for kk in range(8):
        e(f"v_xor_b32 {vr(T0)}, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(KA[kk])}, {vr(T0)}, 5, %[kab]")
        if KBASE:
            e(f"v_add_u32 {vr(KA[kk])}, {KBASE}, {vr(KA[kk])}")
    for j in range(4):
        e(f"v_xor_b32 {vr(T0)}, {j}, %[yh]")
        e(f"v_lshl_add_u32 {vr(VA[j])}, {vr(T0)}, 5, %[vab]")
    e(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
    for qb in range(2):
        e(f"v_mov_b32 {vr(M_RUN[qb])}, 0")
        e(f"v_mov_b32 {vr(L_A[qb])}, 0")
        e(f"v_mov_b32 {vr(L_B[qb])}, 0")
        for r in range(16):
            e(f"v_mov_b32 {vr(MI(qb) + r)}, 0")
    for i in range(128):
        e(f"v_accvgpr_write_b32 {ar(i)}, 0")
    # first tiles: K(0), V(0), K(1) (and K(2) with the three-slot ring)
    n_dma = 0
    for ln in k_dma(0) + v_dma(0) + k_dma(KSLOT) + (k_dma(2 * KSLOT) if k3 else []):
        e(ln)
        n_dma += ln.startswith("buffer_load")
    e(f"s_waitcnt vmcnt({n_dma})")                              # the 16 Q loads (issued first) have landed
    for i in range(64):
        src = 32 + i
        e(f"v_lshlrev_b32 {vr(T0)}, 16, {vr(src)}")
        e(f"v_and_b32 {vr(T1)}, 0xffff0000, {vr(src)}")
        e(f"v_mul_f32 {vr(T0)}, %[sc], {vr(T0)}")
        e(f"v_mul_f32 {vr(T1)}, %[sc], {vr(T1)}")
        e(f"v_cvt_pk_bf16_f32 {vr(T2)}, {vr(T0)}, {vr(T1)}")
        e(f"v_accvgpr_write_b32 {ar(128 + i)}, {vr(T2)}")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e("s_nop 7")
    # ---------------- scores of tile 0 -> set 0 (C = 0)
    ops = weave(qk_mfmas(0, c_zero=True), [qk_read_plan(0)], pre=kread_ops(0, 0) + kread_ops(1, 0))
    pending = linearize(e, ops, [])
    assert not pending
    if k3:
        # K ring bookkeeping: tile t reads slot (t+1) % 3 and refills slot t % 3; the address registers now move to slot 1
        e(f"s_mov_b32 s98, {KSLOT}")                            # slot offset the K addresses point at
        e("s_mov_b32 s99, 0")                                   # slot offset the next K DMA fills
        for kk in range(8):
            e(f"v_add_u32 {vr(KA[kk])}, {KSLOT}, {vr(KA[kk])}")
        for op in kread_ops(0, 0) + kread_ops(1, 0):
            e(op[2])
        pending = ["K0.0", "K0.1", "K1.0", "K1.1"]
    e("s_nop 7")
    e("s_nop 7")                                               # MFMA write of the scores -> VALU
    nomask0 = e.lab("nomask_first")
    e("s_cmp_ge_i32 %[srem], 64")
    e(f"s_cbranch_scc1 {nomask0}")
    mask_block(e, 0)
    e.label(nomask0)
    for ln in rowmax_lines(0):
        e(ln)
    check_and_rescale(e, 0, "first", first=True)
    e("s_mov_b32 s94, 0")                                      # t
    LOOP_PENDING = list(pending)

    LOOP, LAST, EPI = e.lab("loop"), [e.lab("last0"), e.lab("last1")], e.lab("epi")

    def body(p):
        cur, nxt = p, p ^ 1
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        if k3:
            dma = k_dma("s99") + v_dma(nxt)
            kbase = 0                                          # the address registers carry the slot
        else:
            dma = k_dma(p * KSLOT) + v_dma(nxt)
            kbase = nxt * KSLOT
        vbase = VBASE + p * KSLOT
        # ---- phase 1
        mf1 = qk_mfmas(nxt)
        plans = [qk_read_plan(kbase)]
        plans.append(spread(X(dma), 32, 0, 30) if dma_spread else spread(X(dma), 32, 0, 6))
        plans.append(spread(X(softmax_lines(cur, abl_poly)), 32, 2, 32))
        pre = [] if k3 else kread_ops(0, kbase) + kread_ops(1, kbase)
        if vpre:
            vp = [[] for _ in range(32)]
            vp[29], vp[30], vp[31] = [vread_op(0, vbase)], [vread_op(1, vbase)], [vread_op(2, vbase)]
            plans.append(vp)
        ops1 = weave(mf1, plans, pre=pre)
        mark = len(e.lines)
        pend = linearize(e, ops1, LOOP_PENDING)
        # ---- between the phases: mask the new scores if tile t+1 is the last one and partial
        e("s_sub_u32 %[srem], %[srem], 64")                    # keys left from tile t+1 on
        nomask = e.lab(f"nomask_{p}")
        e("s_cmp_ge_i32 %[srem], 64")
        e(f"s_cbranch_scc1 {nomask}")
        e("s_nop 7")
        e("s_nop 7")
        mask_block(e, nxt)
        e.label(nomask)
        # ---- phase 2
        mf2 = pv_mfmas(cur)
        plans = [pv_read_plan(vbase)]
        tail = ([] if abl_nosum else rowsum_lines(cur)) + ([] if abl_nomax else rowmax_lines(nxt))
        if k3:
            # the K addresses step to the next slot once the reads of phase 1 are all issued; the first fragments of
            # the next tile's K are read under the last MFMAs
            step = ["s_mov_b32 s91, s98"] + rotate("s98", 3) + ["s_sub_u32 s91, s98, s91"] + rotate("s99", 3)
            step += [f"v_add_u32 {vr(KA[kk])}, s91, {vr(KA[kk])}" for kk in range(8)]
            plans.append(spread(X(step), 32, 0, 12))
            kp = [[] for _ in range(32)]
            kp[27], kp[29] = kread_ops(0, 0), kread_ops(1, 0)
            plans.append(kp)
        plans.append(spread(X(tail), 32, 1, 32))
        pre = [] if vpre else [vread_op(0, vbase), vread_op(1, vbase), vread_op(2, vbase)]
        pend = linearize(e, weave(mf2, plans, pre=pre), pend)
        assert pend == LOOP_PENDING, (pend, LOOP_PENDING)
        if not abl_nomax:
            check_and_rescale(e, nxt, f"b{p}")
        e("s_add_u32 s94, s94, 1")

    def last(p):
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        vbase = VBASE + p * KSLOT
        for ln in softmax_lines(p):
            e(ln)
        ops2 = weave(pv_mfmas(p), [pv_read_plan(vbase), spread(X(rowsum_lines(p)), 32, 1, 32)],
                     pre=[vread_op(0, vbase), vread_op(1, vbase), vread_op(2, vbase)])
        linearize(e, ops2, LOOP_PENDING)

    e.label(LOOP)
    for p in range(2):
        e("s_cmp_eq_u32 s94, %[slast]")
        e(f"s_cbranch_scc1 {LAST[p]}")
        body(p)
    e(f"s_branch {LOOP}")
    for p in range(2):
        e.label(LAST[p])
        last(p)
        if p == 0:
            e(f"s_branch {EPI}")
    e.label(EPI)

    # ---------------- epilogue: l = sum over both half-waves, O / l -> bf16 -> global
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    for qb in range(2):
        e(f"v_add_f32 {vr(L_A[qb])}, {vr(L_A[qb])}, {vr(L_B[qb])}")
    for qb in range(2):
        e(f"v_mov_b32 {vr(L_B[qb])}, {vr(L_A[qb])}")
    e("s_nop 1")
    for qb in range(2):
        e(f"v_permlane32_swap_b32 {vr(L_A[qb])}, {vr(L_B[qb])}")
    for qb in range(2):
        e(f"v_add_f32 {vr(L_A[qb])}, {vr(L_A[qb])}, {vr(L_B[qb])}")
        e(f"v_rcp_f32 {vr(INV[qb])}, {vr(L_A[qb])}")
    for qb in range(2):
        so = "0" if qb == 0 else "%[so1]"
        for db in range(4):
            for g in range(4):
                regs = [O(qb, db) + 4 * g + i for i in range(4)]
                tmp = [E0 + 4 * (g & 1) + i for i in range(4)]
                for a_, t_ in zip(regs, tmp):
                    e(f"v_accvgpr_read_b32 {vr(t_)}, {ar(a_)}")
                for t_ in tmp:
                    e(f"v_mul_f32 {vr(t_)}, {vr(t_)}, {vr(INV[qb])}")
                if partial:
                    e(f"buffer_store_dwordx4 {vr(tmp[0], 4)}, %[voo], %[ro], {so} offen offset:{db * 128 + g * 32}")
                    e("s_nop 1")
                    continue
                e(f"v_cvt_pk_bf16_f32 {vr(tmp[0])}, {vr(tmp[0])}, {vr(tmp[1])}")
                e(f"v_cvt_pk_bf16_f32 {vr(tmp[1])}, {vr(tmp[2])}, {vr(tmp[3])}")
                e(f"buffer_store_dwordx2 {vr(tmp[0], 2)}, %[voo], %[ro], {so} offen offset:{db * 64 + g * 16}")
    # log-sum-exp (natural log) for the backward, lower half-wave.  No lse requested: the descriptor %[rl] has zero
    # records and the two stores are dropped by the hardware
    for qb in range(2):
        e(f"v_log_f32 {vr(T0 + qb)}, {vr(L_A[qb])}")
    e("s_nop 0")
    for qb in range(2):
        e(f"v_add_f32 {vr(T0 + qb)}, {vr(T0 + qb)}, {vr(M_RUN[qb])}")
        e(f"v_mul_f32 {vr(T0 + qb)}, 0x3f317218, {vr(T0 + qb)}")
    e("s_mov_b64 s[96:97], exec")
    e("s_mov_b64 exec, 0xffffffff")
    e(f"buffer_store_dword {vr(T0)}, %[vol], %[rl], 0 offen")
    e(f"buffer_store_dword {vr(T1)}, %[vol], %[rl], 0 offen offset:128")
    e("s_mov_b64 exec, s[96:97]")
    e("s_waitcnt vmcnt(0)")
    return e


N_VARIANTS = 4
LDS_BYTES = {0: 5 * KSLOT if ABL else 4 * KSLOT, 1: 5 * KSLOT if ABL else 4 * KSLOT, 2: 5 * KSLOT, 3: 5 * KSLOT}
CLOBBER_V = range(12, 256)
CLOBBER_A = range(0, 256)
CLOBBER_S = range(91, 100)


def main():
    print("// GENERATED by gen_attn_w64.py — do not edit; edit the generator.")
    # Round 5: the shipped library carries the default stream (V2) and the split-KV workers' stream (V3) only; V0 / V1
    # (the steps that led to V2, kept side by side for A/B timing in rounds 2-4) are emitted for the timing-only ablation
    # builds (OMH_ATTN_ABL), which re-use their slots.
    for v in (range(N_VARIANTS) if ABL else (2, 3)):
        e = generate(v)
        print(f"#define OMH_ATTN_W64_ASM_V{v} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        print(f"#define OMH_ATTN_W64_LDS_V{v} {LDS_BYTES[v]}")
        n_mfma = sum("v_mfma" in ln for ln in e.lines)
        print(f"// variant {v}: {len(e.lines)} lines, {n_mfma} MFMA")
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in CLOBBER_S] + [f'"v{i}"' for i in CLOBBER_V] + \
           [f'"a{i}"' for i in CLOBBER_A]
    print("#define OMH_ATTN_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))
    print(f"#define OMH_ATTN_W64_VARIANTS {N_VARIANTS}")


if __name__ == "__main__":
    main()
