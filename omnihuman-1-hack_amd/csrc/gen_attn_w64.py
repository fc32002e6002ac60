#!/usr/bin/env python3
"""Generator of the instruction stream of ``flash_attn_fwd_d128_w64_kernel`` (attention_w64.hip).

The long-sequence attention kernel of round 2: 4 waves x 64 query rows per workgroup, ONE wave per SIMD owning
the whole 512-entry register file.  hipcc cannot be made to keep 128 accumulators + 64 operand registers in the
accumulator half without shuffling them through v_accvgpr_read/write (round 1, DESIGN.md 4.4 last row: 744 TF), so
the main loop is written out instruction by instruction here, with a fixed register map, and pasted into the
kernel as one ``asm volatile`` block (``attention_w64_asm.inc``).  The surrounding HIP code computes addresses
and descriptors only.

    python gen_attn_w64.py > attention_w64_asm.inc        (build.py does this when the generator changed)

Math and layouts are those of attention.hip (S^T = K Q^T, O^T = V^T P^T, key bits 2<->3 swapped so P never moves
between lanes; K [64][128] / V^T [128][64] bf16 tiles, 16-byte-slot XOR swizzle, LDS-DMA staged).  What is new:

  * two 32-query blocks per wave share every K / V^T fragment read: 0.5 ds_read_b128 per MFMA instead of 1;
  * O (128 regs) and the pre-scaled Q fragments (64 regs) live in AGPRs, K / V^T fragment rings too — the arch
    VGPRs hold two score sets (128), the packed P (32) and the softmax state;
  * Q is multiplied by scale*log2(e) once (bf16), and the running max enters the scores through the MFMA's C
    operand (16 registers per query block holding -m): p = exp2(S) needs NO per-element subtract / fma;
  * the running max is only moved when a row's new maximum exceeds it by more than THR = 4 (factor 16 in P):
    the rescale of O (AGPR round trip) is a rare slow path, never on the hot path;
  * per tile: phase 1 = K(t+1).Q^T (32 MFMA) || exp2 + bf16 packing of tile t (96 VALU) || K fragment reads,
    phase 2 = V^T(t).P^T(t) (32 MFMA) || row sums of tile t + row max of tile t+1 (~100 VALU) || V^T fragment reads;
    <= 4.2 non-MFMA issues per MFMA (the budget is 5: MI355X_MICROARCH.md, one wave per SIMD).

Register map (asm-owned; inputs stay in the compiler's operand registers v0..v31 / SGPRs):
  a[0:127]    O^T accumulators   [qb][db] 16 each
  a[128:191]  Q fragments        [qb][kk] 4 each (bf16x8)
  a[192:215]  K fragment ring    3 stages x [kb] x 4
  a[216:231]  V^T fragment ring  4 stages x 4
  v[32:95]    score set 0        [qb][kb] 16 each          v[96:159] score set 1
  v[160:191]  MI = -m_run        [qb] 16 copies (MFMA C operand)
  v[192:223]  P packed bf16      [qb][kb][a] 4 each
  v[224:255]  state / temporaries
"""
import sys

THR = "4.0"                      # inline constant: rescale when a row max exceeds the running max by > 4 (log2 units)

# ---------------------------------------------------------------- register map
def O(qb, db):      return (qb * 4 + db) * 16
def Q(qb, kk):      return 128 + (qb * 8 + kk) * 4
def KR(stage, kb):  return 192 + stage * 8 + kb * 4
def VR(stage):      return 216 + stage * 4
def S(st, qb, kb):  return 32 + st * 64 + (qb * 2 + kb) * 16
def MI(qb):         return 160 + qb * 16
def P(qb, kb, a):   return 192 + ((qb * 2 + kb) * 2 + a) * 4

KA = list(range(12, 20))            # K fragment LDS addresses per kk (computed in the prologue)
VA = list(range(20, 24))            # V^T fragment LDS addresses per (kb, a)
# scratch SGPRs (named in the clobber list): s92 / s93 DMA offsets, s94 tile counter, s95 K tile stride, s[96:97] exec

M_RUN = [224, 225]
L_A = [226, 227]
L_B = [228, 229]
MXA = [230, 231]
MXB = [232, 233]
T0, T1, T2, T3 = 234, 235, 236, 237
NEGINF = 238
VLIM = 239
INV = [240, 241]
E0 = 242                          # 242..249 epilogue scratch (8)


def vr(lo, n=1):   return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"
def ar(lo, n=1):   return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}:")

    def text(self):
        out = []
        for ln in self.lines:
            out.append('    "%s\\n\\t"' % ln)
        return "\n".join(out)


def lab(name):
    return f".Lw64_{name}_%="


# ---------------------------------------------------------------- phase building blocks
def interleave(e, mfmas, fillers_per_gap, pre=None):
    """mfmas: list of (wait_or_None, text).  fillers_per_gap: list (len == len(mfmas)) of lists of filler lines issued
    AFTER the MFMA of that gap.  pre: lines before the first MFMA."""
    for ln in pre or []:
        e(ln)
    for (wait, txt), fill in zip(mfmas, fillers_per_gap):
        if wait is not None:
            e(wait)
        e(txt)
        for ln in fill:
            e(ln)


def spread(items, gaps, start=0, end=None, per_gap_cap=None):
    """Distribute `items` (ordered) over gaps[start:end] as evenly as possible; returns list of lists."""
    end = gaps if end is None else end
    n = end - start
    out = [[] for _ in range(gaps)]
    if not items:
        return out
    for i, it in enumerate(items):
        g = start + min(n - 1, (i * n) // len(items))
        out[g].append(it)
    return out


def merge(*plans):
    gaps = len(plans[0])
    return [sum((p[g] for p in plans), []) for g in range(gaps)]


def qk_phase(e, nxt, kslot, cur, with_softmax, dma_lines, first_tile_c_zero=False):
    """K(t+1).Q^T -> score set `nxt` (C = MI, i.e. scores - m_run), K fragments from ring slot `kslot`;
    under it: exp2 + bf16 packing of score set `cur` (if with_softmax) and the LDS-DMA issue of the next tiles."""
    kbase = kslot * 16384
    reads = []                      # (line) in issue order: step kk -> 2 reads (kb 0, 1)

    def kread(kk):
        st = kk % 3
        return [f"ds_read_b128 {ar(KR(st, 0), 4)}, {vr(KA[kk])} offset:{kbase}",
                f"ds_read_b128 {ar(KR(st, 1), 4)}, {vr(KA[kk])} offset:{kbase + 8192}"]

    # MFMA list: per kk: (kb0,qb0) (kb0,qb1) (kb1,qb0) (kb1,qb1)
    mf = []
    outstanding = 0                 # ds_reads issued and not yet known complete, in order
    # reads for kk = 0, 1 go in front
    pre = kread(0) + kread(1)
    issued = 4                      # reads issued so far
    fill_reads = [[] for _ in range(32)]
    for kk in range(8):
        for j, (kb, qb) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            g = kk * 4 + j
            st = kk % 3
            c = vr(MI(qb), 16) if kk == 0 else vr(S(nxt, qb, kb), 16)
            if kk == 0 and first_tile_c_zero:
                c = "0"
            txt = f"v_mfma_f32_32x32x16_bf16 {vr(S(nxt, qb, kb), 16)}, {ar(KR(st, kb), 4)}, {ar(Q(qb, kk), 4)}, {c}"
            wait = None
            if j == 0 or j == 2:
                # need read index (2kk + kb) complete: outstanding allowed = issued - (2kk + kb + 1)
                need = 2 * kk + kb + 1
                wait = f"s_waitcnt lgkmcnt({issued - need})"
            mf.append((wait, txt))
            # after the LAST MFMA using stage (kk % 3) ... the ring has 3 stages: stage of kk+2 == stage of kk-1, free
            # once step kk-1's MFMAs are issued; issue the reads of step kk+2 after the 2nd MFMA of step kk
            if j == 1 and kk + 2 < 8:
                fill_reads[g] = kread(kk + 2)
                issued += 2
    # softmax of the current tile: 64 exp (in place) + 32 cvt
    valu = []
    if with_softmax:
        for qb in range(2):
            for kb in range(2):
                base = S(cur, qb, kb)
                for a in range(2):
                    for eidx in range(4):
                        r0 = base + 8 * a + 2 * eidx
                        valu.append(f"v_exp_f32 {vr(r0)}, {vr(r0)}")
                        valu.append(f"v_exp_f32 {vr(r0 + 1)}, {vr(r0 + 1)}")
                # packing trails the exponentials by one (qb, kb) block: no trans -> use adjacency
                for a in range(2):
                    for eidx in range(4):
                        r0 = base + 8 * a + 2 * eidx
                        valu.append(("cvt", f"v_cvt_pk_bf16_f32 {vr(P(qb, kb, a) + eidx)}, {vr(r0)}, {vr(r0 + 1)}"))
        # reorder: keep each block's cvts a few instructions behind its exps (they are: 16 exps precede them)
        valu = [v[1] if isinstance(v, tuple) else v for v in valu]
    plan_valu = spread(valu, 32, 2, 32)
    plan_dma = spread(dma_lines, 32, 0, 6)
    interleave(e, mf, merge(plan_dma, fill_reads, plan_valu), pre=pre)


def softmax_only(e, cur):
    for qb in range(2):
        for kb in range(2):
            base = S(cur, qb, kb)
            for r in range(16):
                e(f"v_exp_f32 {vr(base + r)}, {vr(base + r)}")
            for a in range(2):
                for eidx in range(4):
                    r0 = base + 8 * a + 2 * eidx
                    e(f"v_cvt_pk_bf16_f32 {vr(P(qb, kb, a) + eidx)}, {vr(r0)}, {vr(r0 + 1)}")


def rowsum_lines(cur):
    out = []
    for qb in range(2):
        regs = [S(cur, qb, kb) + r for kb in range(2) for r in range(16)]
        for i, r in enumerate(regs):
            acc = L_A[qb] if i % 2 == 0 else L_B[qb]
            out.append(f"v_add_f32 {vr(acc)}, {vr(acc)}, {vr(r)}")
    # interleave the two query blocks so that dependent adds are 4 apart
    a, b = out[:32], out[32:]
    mixed = []
    for i in range(0, 32, 2):
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


def pv_phase(e, cur, vslot, tail_valu):
    """O^T += V^T(t) P^T(t): 16 fragment steps x 2 query blocks; V^T fragments from ring slot `vslot` (ring of 4,
    read 3 steps ahead); `tail_valu` lines are spread under the MFMAs."""
    vbase = 32768 + vslot * 16384

    def vread(i):
        kb, a, db = i >> 3, (i >> 2) & 1, i & 3
        return f"ds_read_b128 {ar(VR(i % 4), 4)}, {vr(VA[2 * kb + a])} offset:{vbase + db * 4096}"

    pre = [vread(0), vread(1), vread(2)]
    issued = 3
    mf, fill_reads = [], [[] for _ in range(32)]
    for i in range(16):
        kb, a, db = i >> 3, (i >> 2) & 1, i & 3
        for qb in range(2):
            g = 2 * i + qb
            wait = f"s_waitcnt lgkmcnt({issued - (i + 1)})" if qb == 0 else None
            mf.append((wait, f"v_mfma_f32_32x32x16_bf16 {ar(O(qb, db), 16)}, {ar(VR(i % 4), 4)}, "
                             f"{vr(P(qb, kb, a), 4)}, {ar(O(qb, db), 16)}"))
            # stage (i+3)%4 == (i-1)%4 is free once step i-1's MFMAs are issued: read step i+3 after the 1st MFMA of step i
            if qb == 0 and i + 3 < 16:
                fill_reads[g] = [vread(i + 3)]
                issued += 1
    plan = spread(tail_valu, 32, 1, 32)
    interleave(e, mf, merge(fill_reads, plan), pre=pre)


def dma_lines(kslot, vslot):
    """LDS-DMA of K(next-next) -> K ring slot kslot and V^T(next) -> V ring slot vslot: 4 + 4 pieces of 1 KiB per wave.
    Source offsets: per-lane voffset + running scalar offsets (%[skn] / %[svn] hold the tile base, pieces add a
    multiple of the piece stride held in %[skp] / %[svp])."""
    out = []
    for j in range(4):
        out.append(f"s_add_u32 m0, %[ldsw], {kslot * 16384 + j * 4096}")
        if j == 0:
            out.append("s_mov_b32 s92, %[skn]")
        else:
            out.append("s_add_u32 s92, s92, %[skp]")
        out.append("buffer_load_dwordx4 %[vok], %[rk], s92 offen lds")
    for j in range(4):
        out.append(f"s_add_u32 m0, %[ldsw], {32768 + vslot * 16384 + j * 4096}")
        if j == 0:
            out.append("s_mov_b32 s93, %[svn]")
        else:
            out.append("s_add_u32 s93, s93, %[svp]")
        out.append("buffer_load_dwordx4 %[vov], %[rv], s93 offen lds")
    out.append("s_add_u32 %[skn], %[skn], s95")
    out.append("s_add_u32 %[svn], %[svn], 128")
    return out


def mask_block(e, st, tag):
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
    d, al, t = T0, T1, T2
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
    """After the row max of set `st` is in MXA: per query block, branch to the rescale when needed."""
    for qb in range(2):
        skip = lab(f"nores_{tag}_{qb}")
        if not first:
            e(f"v_cmp_lt_f32 vcc, {THR}, {vr(MXA[qb])}")
            e(f"s_cbranch_vccz {skip}")
        slow_path(e, st, qb, first)
        if not first:
            e.label(skip)


# ---------------------------------------------------------------- whole stream
def generate():
    e = Emit()
    # ---------------- prologue: Q fragments -> pre-scaled bf16 -> AGPRs
    for qb in range(2):
        for kk in range(8):
            so = "0" if qb == 0 else "%[sq1]"
            e(f"buffer_load_dwordx4 {vr(32 + (qb * 8 + kk) * 4, 4)}, %[voq], %[rq], {so} offen offset:{kk * 32}")
    # constants / state while the loads fly
    e("s_lshl_b32 s95, %[skp], 2")
    for kk in range(8):
        e(f"v_xor_b32 {vr(T0)}, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(KA[kk])}, {vr(T0)}, 5, %[kab]")
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
    # first tiles' DMA: K(0) -> K slot 0, V(0) -> V slot 0 ; then K(1) -> K slot 1 (V slot unused: point it at slot 1)
    for ln in dma_lines(0, 0):
        e(ln)
    # dma_lines advanced %[skn] to tile 1 and %[svn] to tile 1; issue K(1) only
    for j in range(4):
        e(f"s_add_u32 m0, %[ldsw], {16384 + j * 4096}")
        if j == 0:
            e("s_mov_b32 s92, %[skn]")
        else:
            e("s_add_u32 s92, s92, %[skp]")
        e("buffer_load_dwordx4 %[vok], %[rk], s92 offen lds")
    e("s_add_u32 %[skn], %[skn], s95")                  # -> tile 2
    # Q: wait for the 16 loads (the 12 DMAs behind them stay in flight): vmcnt counts in order
    e("s_waitcnt vmcnt(12)")
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
    # ---------------- scores of tile 0 -> set 0 (C = 0), no softmax under it
    qk_phase(e, nxt=0, kslot=0, cur=1, with_softmax=False, dma_lines=[], first_tile_c_zero=True)
    e("s_nop 7")
    e("s_nop 7")                                               # MFMA write of the scores -> VALU
    # mask (only when tile 0 is partial), row max, forced first placement of the running max
    nomask0 = lab("nomask_first")
    e("s_cmp_ge_i32 %[srem], 64")
    e(f"s_cbranch_scc1 {nomask0}")
    mask_block(e, 0, "first")
    e.label(nomask0)
    for ln in rowmax_lines(0):
        e(ln)
    check_and_rescale(e, 0, "first", first=True)
    e("s_mov_b32 s94, 0")                                   # t

    # ---------------- main loop, two tiles per trip
    LOOP, LAST = lab("loop"), [lab("last0"), lab("last1")]
    EPI = lab("epi")

    def body(p):
        cur, nxt = p, p ^ 1
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        # phase 1: K(t+1).Q^T -> set nxt from K slot (t+1)&1 = nxt ; DMA K(t+2) -> K slot p, V(t+1) -> V slot nxt
        qk_phase(e, nxt=nxt, kslot=nxt, cur=cur, with_softmax=True, dma_lines=dma_lines(p, nxt))
        # between the phases: mask the new scores if tile t+1 is the last one and partial
        e("s_sub_u32 %[srem], %[srem], 64")                    # keys left from tile t+1 on
        nomask = lab(f"nomask_{p}")
        e("s_cmp_ge_i32 %[srem], 64")
        e(f"s_cbranch_scc1 {nomask}")
        e("s_nop 7")
        e("s_nop 7")
        mask_block(e, nxt, f"b{p}")
        e.label(nomask)
        # phase 2: P.V of tile t ; row sums of tile t, then row max of tile t+1 (its MFMAs are >= 16 MFMAs back)
        pv_phase(e, cur=cur, vslot=p, tail_valu=rowsum_lines(cur) + rowmax_lines(nxt))
        check_and_rescale(e, nxt, f"b{p}")
        e("s_add_u32 s94, s94, 1")

    def last(p):
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        softmax_only(e, p)
        pv_phase(e, cur=p, vslot=p, tail_valu=rowsum_lines(p))

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
                e(f"v_cvt_pk_bf16_f32 {vr(tmp[0])}, {vr(tmp[0])}, {vr(tmp[1])}")
                e(f"v_cvt_pk_bf16_f32 {vr(tmp[1])}, {vr(tmp[2])}, {vr(tmp[3])}")
                e(f"buffer_store_dwordx2 {vr(tmp[0], 2)}, %[voo], %[ro], {so} offen offset:{db * 64 + g * 16}")
                if g & 1:
                    e("s_nop 0")
    # log-sum-exp (natural log) for the backward: lanes of the lower half-wave, when requested
    # (no lse requested: the descriptor %[rl] has zero records and the two stores are dropped by the hardware)
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


CLOBBER_V = range(12, 256)
CLOBBER_A = range(0, 256)


def main():
    e = generate()
    print("// GENERATED by gen_attn_w64.py — do not edit; edit the generator.")
    print("#define OMH_ATTN_W64_ASM \\")
    body = e.text().split("\n")
    print(" \\\n".join(body))
    print("")
    clob = ['"memory"', '"vcc"', '"scc"', '"s92"', '"s93"', '"s94"', '"s95"', '"s96"', '"s97"'] + [f'"v{i}"' for i in CLOBBER_V] + [f'"a{i}"' for i in CLOBBER_A]
    print("#define OMH_ATTN_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print(", \\\n    ".join(rows).join(["    ", ""]))
    n_mfma = sum("v_mfma" in ln for ln in e.lines)
    print(f"// {len(e.lines)} lines, {n_mfma} MFMA")


if __name__ == "__main__":
    main()
