#!/usr/bin/env python3
"""Generator of the instruction streams of ``gemm_bf16_nt_w64_kernel<EPI>`` (gemm_w64.hip): C = epi(A B^T) for the
large DiT GEMMs on a 256(m) x 384(n) x 64 workgroup tile — 4 waves, ONE per SIMD, each owning a 128 x 192 patch with
384 fp32 accumulators (256 in AGPRs + 128 in arch VGPRs).

Why: the 8-wave 256 x 256 kernel (gemm_bf16.hip) is bound by the L2 -> LDS fill stream (22 B/clk/CU);
what is left is bytes per flop, and a 256 x 384 tile moves 17 % fewer (153.6 flop per staged byte against 128).  It
does not fit two waves per SIMD (192 accumulators + fragments per wave > 256), and a compiler-scheduled one-wave
variant lost to its own LDS-DMA issue stalls in round 1 — hence a generated stream, as for the attention kernel
(gen_attn_w64.py): fixed register map, exact lgkmcnt waits, the 20 DMA pieces of a k step spread between the MFMAs.

    python gen_gemm_w64.py > gemm_w64_asm.inc

Main loop (as gemm_bf16.hip): operand tiles [rows][64] bf16 in LDS, 16-byte-slot XOR swizzle ((row >> 1) & 7) applied on
the LDS-DMA source address and on the ds_read_b128; weights (B, n) in the MFMA A slot, activations (A, m) in the B
slot, so a lane ends up with 4 consecutive n of one row m.  Two stages of 80 KiB (X tile 32 KiB | W tile 48 KiB).

Epilogues (one stream per kind): "f32" C = acc + bias, "bf16" C = bf16(acc + bias), "gelu" C = bf16(gelu_tanh(acc +
bias)), "resid" C fp32 += (acc + bias) * gate[m, n] (model.py:296,313,328).  A lane's four register quads of an
accumulator tile are widened to two runs of 8 consecutive n with v_permlane32_swap (cdna_hip_programming.md T21), so
every global access is 16 / 32 contiguous bytes per lane; the per-column vectors (bias, gate, bias*gate) are built
once per wave in LDS (free after the k loop) and read back as ds_read_b128; columns >= N are masked with EXEC, rows
>= M are dropped by the buffer descriptors.

Register map: a[0:255] accumulator tiles 0..15, v[128:255] tiles 16..23 (tile = 4 i + j, i = n tile, j = m tile);
k loop: v[32:71] / v[72:111] fragment buffers, v[12:19] X fragment addresses [stage][kk], v[20:27] W addresses;
epilogue: v[32:47] tile values, v[48:79] column vectors, v[80:127] prefetched C rows, v12..v31 misc;
s[60:75] unpacked scalar arguments, s[80:91] scratch.
"""
import os
import sys

NI, NJ = 6, 4                   # n tiles (weights) x m tiles (activations) per wave
STAGE = 81920
WOFF = 32768                    # W tile behind the X tile inside a stage
FB = 40                         # registers of one fragment buffer: 4 NI + 4 NJ
NW = 12                         # W pieces (1 KiB) per wave and k tile: 2 NI
KINDS = ("f32", "bf16", "gelu", "resid")


PGR = False                     # configure(4, pgr=True): the "p256" streams (main_loop_pgr)
SET = (96, 160)                 # p256: v[96:159] / v[160:223] = the 16 pieces (4 registers each) of a k tile in flight
VDW = (224, 225)                # p256: LDS write address of the lane in stage 0 / 1 (X region; W: + WOFF)


def configure(ni, pgr=False):
    """Tile configuration of the streams generated next: ni = 6 -> 256 x 384 (the default), ni = 3 -> 256 x 192 (the
    "resid192" stream: 12 accumulator tiles per wave, all in AGPRs; v[96:255] + a[192:223] hold the tile's OLD C values,
    requested during the k loop), ni = 4 with ``pgr`` -> 256 x 256, all 16 accumulator tiles in AGPRs and the 128 free
    registers holding two k tiles of operands IN FLIGHT (main_loop_pgr)."""
    global NI, STAGE, FB, NW, NTILES, PGR
    PGR = pgr
    NI = ni
    STAGE = 32768 + ni * 8192
    FB = 4 * ni + 4 * NJ
    NW = 2 * ni
    NTILES = NI * NJ

# fixed scalar registers (unpacked from the 64-bit operand pairs in the prologue)
S_LDX, S_LDW, S_SXB, S_SWB, S_SXS, S_SWS, S_NK, S_SCB = "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67"
S_SCJ, S_N, S_MB, S_G1LO, S_G1ST, S_GC, S_COL0, S_REG = "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75"
S_NXB, S_NWB, S_NEXT = "s76", "s77", "s78"      # next tile of this workgroup: X / W source offsets, != 0 if there is one
S_COLN = "s89"                  # first column of the wave (elements) = S_COL0 / 4
NPAIRS = 10                     # p0..p9 -> s[60:79]


def acc(i, j):
    t = i * NJ + j
    return (f"a[{t * 16}:{t * 16 + 15}]") if t < 16 else (f"v[{128 + (t - 16) * 16}:{128 + (t - 16) * 16 + 15}]")


def wfrag(buf, i):   return 32 + buf * FB + 4 * i
def xfrag(buf, j):   return 32 + buf * FB + 4 * NI + 4 * j
def XA(s, kk):       return 12 + s * 4 + kk
def WA(s, kk):       return 20 + s * 4 + kk
def vr(lo, n=1):     return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


class Emit:
    def __init__(self, tag):
        self.lines, self.tag = [], tag

    def __call__(self, s):
        self.lines.append(s)

    def lab(self, name):
        return f".Lgw64{self.tag}_{name}_%="

    def label(self, name):
        self.lines.append(f"{name}:")

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


def linearize(e, ops, pending):
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
                assert allowed <= 15, allowed
                e(f"s_waitcnt lgkmcnt({allowed})")
                pending = pending[last + 1:]
            e(op[1])
        else:
            e(op[1])
    return pending


def frag_reads(stage, kk, buf):
    out = []
    for i in range(NI):
        out.append(("r", f"W{buf}.{i}", f"ds_read_b128 {vr(wfrag(buf, i), 4)}, {vr(WA(stage, kk))} offset:{i * 4096}"))
    for j in range(NJ):
        out.append(("r", f"X{buf}.{j}", f"ds_read_b128 {vr(xfrag(buf, j), 4)}, {vr(XA(stage, kk))} offset:{j * 4096}"))
    return out


SWAP = False                    # "bf16vt": activations in the MFMA A slot, weights in the B slot -> a lane holds 4 consecutive m


M16 = False                     # timing-only ablation: every 32x32x16 MFMA as TWO v_mfma_f32_16x16x32_bf16 on the same operand
#                                 registers (the same flops, LDS and global traffic; the results are garbage) — does the shape
#                                 that sustains 2.16 instead of 1.88 PFLOP/s on data-like operands (tools/mfma_shape_probe.py) pay
#                                 inside a real k loop?


def group_mfmas(buf, first=False):
    out = []
    for i in range(NI):
        for j in range(NJ):
            c = "0" if first else acc(i, j)
            a_, b_ = (xfrag(buf, j), wfrag(buf, i)) if SWAP else (wfrag(buf, i), xfrag(buf, j))
            if M16:
                t = i * NJ + j
                base, f = (t * 16, "a") if t < 16 else (128 + (t - 16) * 16, "v")
                for half in range(2):
                    d = f"{f}[{base + 4 * half}:{base + 4 * half + 3}]"
                    out.append(("m", f"v_mfma_f32_16x16x32_bf16 {d}, {vr(a_, 4)}, {vr(b_, 4)}, {d}",
                                [f"W{buf}.{i}", f"X{buf}.{j}"] if half == 0 else []))
                continue
            out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(a_, 4)}, {vr(b_, 4)}, {c}",
                        [f"W{buf}.{i}", f"X{buf}.{j}"]))
    return out


def dma_piece(stage, operand, q, xb=S_SXB, wb=S_SWB):
    """One 1 KiB LDS-DMA piece.  X: wave w fetches rows w*64 + 8 q (q < 8); W: rows w*16 NI + 8 q (q < 2 NI).
    s81 / s82: running source offsets of the X / W piece; s80: k byte offset of the tile being fetched."""
    base = stage * STAGE + (WOFF if operand == "w" else 0)
    vo = ("%[vow1]" if q & 1 else "%[vow0]") if operand == "w" else ("%[vox1]" if q & 1 else "%[vox0]")
    rs = "%[rb]" if operand == "w" else "%[ra]"
    sreg = "s82" if operand == "w" else "s81"
    lds = S_LDW if operand == "w" else S_LDX
    out = [f"s_add_u32 m0, {lds}, {base + q * 1024}"]
    if q == 0:
        out.append(f"s_add_u32 {sreg}, {wb if operand == 'w' else xb}, s80")
    else:
        out.append(f"s_add_u32 {sreg}, {sreg}, {S_SWS if operand == 'w' else S_SXS}")
    out.append(f"buffer_load_dwordx4 {vo}, {rs}, {sreg} offen lds")
    return [("x", ln) for ln in out]


def all_pieces(stage, xb=S_SXB, wb=S_SWB):
    return [dma_piece(stage, "x", q, xb, wb) for q in range(8)] + [dma_piece(stage, "w", q, xb, wb) for q in range(NW)]


def unpack(e, pairs=range(NPAIRS)):
    """Nothing to emit: gemm_w64.hip binds the packed scalar operands to s[60:79] directly ("{s[60:61]}" ...)."""


def tile_prologue(e, xb, wb):
    """k tile 0 -> stage 0 (all 8 + NW pieces per wave), k tile 1 -> stage 1 (the 8 X pieces; the k loop issues the NW
    W pieces): 16 + NW LDS-DMA instructions."""
    e("s_mov_b32 s80, 0")
    for ops in all_pieces(0, xb, wb):
        for op in ops:
            e(op[1])
    e("s_mov_b32 s80, 128")
    for ops in (all_pieces(1, xb, wb) if PGR else all_pieces(1, xb, wb)[:8]):   # p256: k tile 1 whole (its k loop fetches to registers)
        for op in ops:
            e(op[1])


def pro_pieces():
    """LDS-DMA pieces of a tile's prologue that are younger than k tile 0's."""
    return 8 + NW if PGR else 8


def spread_after(mfmas, extras, start=0, end=None):
    end = len(mfmas) if end is None else end
    n = end - start
    slots = [[] for _ in mfmas]
    for k, ex in enumerate(extras):
        slots[start + min(n - 1, (k * n) // max(1, len(extras)))].extend(ex)
    ops = []
    for m, s in zip(mfmas, slots):
        ops.append(m)
        ops.extend(s)
    return ops


SCHED = os.environ.get("OMH_GW64_SCHED", "0")     # experiment knob: where the W pieces of a k step are issued


def with_dma_tail(ops, dm, start=12):
    """Insert the DMA op lists `dm` after every other MFMA from the (start+1)-th on."""
    out, idx, cnt = [], 0, 0
    for op in ops:
        out.append(op)
        if op[0] == "m":
            cnt += 1
            if cnt > start and idx < len(dm) and (cnt - start) % 2 == 1:
                out.extend(dm[idx]); idx += 1
    while idx < len(dm):
        out.extend(dm[idx]); idx += 1
    return out


CPRE = 96                      # resid192: v[96:255] = old C of tiles 0..9, a[192:223] = tiles 10, 11


def c_slot(k, lo, n=4):
    """Register range [lo, lo + n) of accumulator tile k's old C values."""
    base, f = (CPRE + 16 * k, "v") if k < 10 else (192 + 16 * (k - 10), "a")
    return f"{f}[{base + lo}:{base + lo + n - 1}]"


def c_prefetch(k):
    """The four 16-byte pieces per lane of the OLD C values of accumulator tile k (processing order = (i, j) = divmod
    (k, NJ)): issued from k step k of the main loop, behind that step's W pieces, so the step's counted vmcnt leaves them
    in flight for a whole k step."""
    i, j = divmod(k, NJ)
    ops = [("x", f"s_mul_i32 s88, {S_SCJ}, {j}"), ("x", f"s_add_u32 s88, s88, {S_SCB}")]
    for p in range(2):
        for q in range(2):
            lo = p * 8 + q * 4
            dst = c_slot(k, lo)
            ops.append(("x", f"buffer_load_dwordx4 {dst}, %[voc], %[rcin], s88 offen offset:{(i * 32 + 16 * p) * 4 + q * 16}"))
    return ops


def main_loop(e, epi_vmem, cpre=False):
    """The k loop of one output tile.  On entry the tile's prologue DMA (tile_prologue) is in flight, followed in issue
    order by `epi_vmem` stores / loads of the previous tile's epilogue (none for the workgroup's first tile, whose
    prologue stream ended with its own wait).  ``cpre``: k steps 0 .. NTILES-1 are peeled and each requests one
    accumulator tile's old C values (c_prefetch); K >= (NTILES + 2) k tiles."""
    unpack(e)
    for kk in range(4):
        e(f"v_xor_b32 v112, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(XA(0, kk))}, v112, 5, %[xab]")
        e(f"v_lshl_add_u32 {vr(WA(0, kk))}, v112, 5, %[wab]")
        e(f"v_add_u32 {vr(XA(1, kk))}, {STAGE}, {vr(XA(0, kk))}")
        e(f"v_add_u32 {vr(WA(1, kk))}, {STAGE}, {vr(WA(0, kk))}")
    e("s_mov_b32 s80, 128")                                     # k offset of the tile whose W pieces come next
    e(f"s_waitcnt vmcnt({min(63, 8 + epi_vmem)})")               # k tile 0 has landed (in-order counter)
    e("s_barrier")
    pend = linearize(e, frag_reads(0, 0, 0), [])
    LOOP_PENDING = list(pend)
    LOOP, DONE = e.lab("loop"), e.lab("done")

    NM = NI * NJ                                                # MFMAs per 16-wide k group

    def body(s, first=False, mode="full", ctile=None):
        """One k step on stage s.  Groups 0..2: MFMAs || reads of the next group || the W pieces of k tile kt+1
        (into stage s^1).  Then everything in flight is waited for, barrier, and group 3 runs || the 8 X pieces of
        k tile kt+2 (into stage s, free now) || reads of group 0 of stage s^1.  mode "nox": the step before the last
        (no k tile kt+2), "none": the last step (nothing to fetch, nothing to read ahead).  ``ctile``: this step also
        requests the old C values of accumulator tile ctile, in group 2 — younger than the step's W pieces, so the
        counted wait below leaves them in flight until the NEXT step's wait."""
        pend = LOOP_PENDING
        rest = all_pieces(s ^ 1)[8:] if mode != "none" else []
        half = NW // 2
        for kk in range(3):
            mf = group_mfmas(kk & 1, first=(first and kk == 0))
            reads = [[r] for r in frag_reads(s, kk + 1, (kk + 1) & 1)]
            ops = spread_after(mf, reads, 0, min(14, NM))
            if SCHED == "1":                                    # all W pieces in group 0
                if kk == 0:
                    ops = with_dma_tail(ops, rest, 0)
            elif SCHED == "2":                                  # 8 in group 0 from the first MFMA, 4 in group 1
                if kk == 0:
                    ops = with_dma_tail(ops, rest[:8], 4)
                elif kk == 1:
                    ops = with_dma_tail(ops, rest[8:], 0)
            elif kk < 2:
                ops = with_dma_tail(ops, rest[kk * half:(kk + 1) * half], 12 if NM >= 24 else 2)
            elif ctile is not None:
                ops = with_dma_tail(ops, [c_prefetch(ctile)], 2)
            pend = linearize(e, ops, pend)
        if mode == "none":
            linearize(e, group_mfmas(1), pend)
            return
        e(f"s_waitcnt vmcnt({4 if ctile is not None else 0})")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        xp = []
        if mode == "full":
            e("s_add_u32 s80, s80, 128")                        # k offset of tile kt+2
            xp = all_pieces(s)[:8]
        reads = [[r] for r in frag_reads(s ^ 1, 0, 0)]
        extras = []
        for k in range(max(len(xp), len(reads))):
            if k < len(reads):
                extras.append(reads[k])
            if k < len(xp):
                extras.append(xp[k])
        pend = linearize(e, spread_after(group_mfmas(1), extras, 0, NM - 2), [])
        assert pend == LOOP_PENDING, (pend, LOOP_PENDING)

    def step_check(tail):
        e(f"s_sub_u32 s83, {S_NK}, s84")                        # k steps left, this one included (>= 2 here)
        e("s_cmp_eq_u32 s83, 2")
        e(f"s_cbranch_scc1 {tail}")

    TAIL1, TAIL0 = e.lab("tail1"), e.lab("tail0")
    if cpre:                                                    # k steps 0 .. NTILES-1, each with one tile's old C
        for kt in range(NTILES):
            body(kt & 1, first=(kt == 0), ctile=kt)
        assert NTILES % 2 == 0
        body(0)                                                 # k step NTILES: the wait that lands the last C tile
        e(f"s_mov_b32 s84, {NTILES + 1}")
    else:
        body(0, first=True)                                     # k step 0 (K >= 3 k tiles: it is never a tail step)
        e("s_mov_b32 s84, 1")
    e.label(LOOP)
    step_check(TAIL1)
    body(1)
    e("s_add_u32 s84, s84, 1")
    step_check(TAIL0)
    body(0)
    e("s_add_u32 s84, s84, 1")
    e(f"s_branch {LOOP}")
    e.label(TAIL1)
    body(1, mode="nox")
    body(0, mode="none")
    e(f"s_branch {DONE}")
    e.label(TAIL0)
    body(0, mode="nox")
    body(1, mode="none")
    e.label(DONE)
    e("s_waitcnt vmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                              # every wave is done with the LDS stages
    for _ in range(3):
        e("s_nop 7")                                            # last MFMA results readable by VALU


VOX, VOW = 226, 234             # p256: v[226:233] / v[234:241] = the lane's source offset of X / W piece q (swizzle + q row blocks)


def load_piece(vset, operand, q):
    """p256: one 1 KiB piece of a k tile as an ordinary buffer load into registers (piece index: X 0..7, W 8..15); the
    same source addresses as dma_piece (swizzle on the lane's offset), the piece's row block in a per-piece offset
    register, the tile's k offset in s81 (X) / s82 (W)."""
    dst = SET[vset] + 4 * (q if operand == "x" else 8 + q)
    vo = (VOW if operand == "w" else VOX) + q
    rs, sreg = ("%[rb]", "s82") if operand == "w" else ("%[ra]", "s81")
    return ("v", f"buffer_load_dwordx4 {vr(dst, 4)}, v{vo}, {rs}, {sreg} offen{LDMOD}")


def write_piece(vset, stage, idx):
    """p256: piece idx of the k tile held in register set vset -> LDS stage `stage`, where the LDS-DMA would have put it
    (lane-linear inside the wave's 1 KiB piece)."""
    off = idx * 1024 if idx < 8 else WOFF + (idx - 8) * 1024
    return ("w", f"P{idx}", f"ds_write_b128 v{VDW[stage]}, {vr(SET[vset] + 4 * idx, 4)} offset:{off}", idx)


LDMOD = ""                      # cache-policy bits of the p256 loads (experiment: " nt", " sc1", ...)
PABL = ""                       # timing-only ablation of main_loop_pgr (OMH_GEMM_W64_P256 = a..e, tools/gemm_p256_abl.py):
#   "noload" no global loads, "nowrite" no LDS writes, "nomem" neither, "nobar" no barrier, "noread" no fragment reads


def linearize_pgr(e, ops, pending, vm):
    """linearize + the in-order VMEM counter.  ``vm`` = [loads outstanding that are OLDER than piece 0 of the k tile
    being written — none — then: loads of that tile and younger ones issued before this op list, loads issued in it]:
    a ds_write of piece i waits until at most (vm[0] - 1 - i) + vm[1] loads are outstanding."""
    pending = list(pending)
    for op in ops:
        if op[0] == "r":
            if PABL == "noread":
                continue
            e(op[2])
            pending.append(op[1])
        elif op[0] == "w":
            if PABL in ("nowrite", "nomem"):
                continue
            if PABL != "noload":
                e(f"s_waitcnt vmcnt({vm[0] - 1 - op[3] + vm[1]})")
            e(op[2])
            pending.append(op[1])
        elif op[0] == "v":
            if PABL in ("noload", "nomem"):
                continue
            e(op[1])
            vm[1] += 1
        elif op[0] == "m":
            need = [t for t in op[2] if t in pending]
            if need:
                last = max(pending.index(t) for t in need)
                allowed = len(pending) - last - 1
                e(f"s_waitcnt lgkmcnt({min(allowed, 15)})")
                pending = pending[last + 1:]
            e(op[1])
        else:
            e(op[1])
    return pending


def slots(mfmas, first_half, second_half):
    """One extra per MFMA gap: ``first_half`` after MFMAs 0.., ``second_half`` after MFMAs 8.. of the group of 16 (a
    single wave per SIMD issues in order: a gap with two or three memory instructions is a gap in which the matrix pipe
    runs dry — measured, tools/gemm_p256_abl.py)."""
    assert len(mfmas) == 16 and len(first_half) <= 8 and len(second_half) <= 8
    ops = []
    for k, m in enumerate(mfmas):
        ops.append(m)
        if k < 8 and k < len(first_half):
            ops.append(first_half[k])
        if k >= 8 and k - 8 < len(second_half):
            ops.append(second_half[k - 8])
    return ops


def main_loop_pgr(e, epi_vmem):
    """The k loop of one 256 x 256 output tile with the operands of two k tiles in flight in REGISTERS.

    The LDS-DMA loop above can only request a k tile when an LDS stage is free for it: 80 of the 160 KiB hold the tile
    being multiplied, so at most one k tile (less, on average) is on its way, and the W pieces of tile kt+1 have half a
    k step to arrive.  Here a k tile is 16 ordinary 1 KiB loads per wave into v[96:159] / v[160:223] (the accumulators
    are all AGPRs), issued TWO steps before the tile is multiplied, and written to the LDS stage one step later, when
    the stage has been read for the last time: every load has a whole k step (>= 2 048 MFMA cycles) to land.  Every gap
    between two MFMAs carries exactly ONE memory instruction (64 MFMAs, 32 fragment reads, 16 loads, 16 LDS writes per
    step).  Step tau on stage s = tau & 1, register set of k tile t = t & 1:
        groups 0..2  MFMAs kk = 0..2 || gaps 0-7: fragment reads of kk + 1 || gaps 8-15: ds_write of k tile tau+1
                     (set s^1 -> stage s^1; 16 in 24 gaps) and the 8 W pieces of k tile tau+2 (-> set s)
        lgkmcnt(0), barrier                      (stage s^1 complete, stage s read for the last time, set s^1 free)
        group 3      MFMAs kk = 3 || gaps 0-7: fragment reads kk = 0 of stage s^1 || gaps 8-15: the 8 X pieces of k tile
                     tau+3 (-> set s^1)
    Same MFMA, same order over k as every other GEMM kernel of the library: identical bits.  The prologue (k tiles 0
    and 1, LDS-DMA, issued from the previous tile's epilogue) and the epilogues are the LDS-DMA streams'."""
    unpack(e)
    for kk in range(4):
        e(f"v_xor_b32 v112, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(XA(0, kk))}, v112, 5, %[xab]")
        e(f"v_lshl_add_u32 {vr(WA(0, kk))}, v112, 5, %[wab]")
        e(f"v_add_u32 {vr(XA(1, kk))}, {STAGE}, {vr(XA(0, kk))}")
        e(f"v_add_u32 {vr(WA(1, kk))}, {STAGE}, {vr(WA(0, kk))}")
    e(f"v_lshl_add_u32 v{VDW[0]}, %[vlane], 4, {S_LDX}")          # the wave's X piece 0 + 16 lane
    e(f"v_add_u32 v{VDW[1]}, {STAGE}, v{VDW[0]}")
    for q in range(8):                                          # per-piece source offsets: row block q of the wave's rows
        e(f"s_mul_i32 s88, {S_SXS}, {q}")
        e(f"v_add_u32 v{VOX + q}, s88, {'%[vox1]' if q & 1 else '%[vox0]'}")
        e(f"s_mul_i32 s88, {S_SWS}, {q}")
        e(f"v_add_u32 v{VOW + q}, s88, {'%[vow1]' if q & 1 else '%[vow0]'}")
    # the X pieces of k tile 2 (what a step's group 3 does for the step after the next)
    e(f"s_add_u32 s81, {S_SXB}, 256")
    n_pre = 0
    if PABL not in ("noload", "nomem"):
        for q in range(8):
            e(load_piece(0, "x", q)[1])
        n_pre = 8
    e(f"s_add_u32 s82, {S_SWB}, 256")                           # W pieces: k tile 2 next
    e(f"s_add_u32 s81, {S_SXB}, 384")                           # X pieces: k tile 3 next
    e(f"s_waitcnt vmcnt({min(63, 16 + epi_vmem + n_pre)})")      # k tile 0 has landed (in-order counter)
    e("s_barrier")
    pend = linearize_pgr(e, frag_reads(0, 0, 0), [], [0, 0])
    LOOP_PENDING = list(pend)
    LOOP, DONE = e.lab("loop"), e.lab("done")

    def body(s, first=False, left=3):
        """``left``: k steps after this one (>= 3: the steady state).  ``first``: step 0 — k tile 1 came by LDS-DMA, there
        is nothing to write."""
        wl = [load_piece(s, "w", q) for q in range(8)] if left >= 2 else []             # W pieces of k tile tau+2
        xl = [load_piece(s ^ 1, "x", q) for q in range(8)] if left >= 3 else []         # X pieces of k tile tau+3
        wr = [write_piece(s ^ 1, s ^ 1, i) for i in range(16)] if (left >= 1 and not first) else []
        # second halves of groups 0..2: W W L, W W L, ... (16 writes, 8 loads in 24 gaps)
        mix = []
        for k in range(8):
            mix += wr[2 * k:2 * k + 2] + wl[k:k + 1]
        mix += [None] * (24 - len(mix))
        # loads older than piece 0 of the k tile being written do not exist; counted from it: its own 16, the 8 X pieces
        # of the next k tile (previous step's group 3)
        vm = [16 + (8 if left >= 2 else 0), 0]
        pend = LOOP_PENDING
        for kk in range(3):
            mf = group_mfmas(kk & 1, first=(first and kk == 0))
            rd = frag_reads(s, kk + 1, (kk + 1) & 1)
            pend = linearize_pgr(e, slots(mf, rd, [m for m in mix[8 * kk:8 * kk + 8] if m is not None]), pend, vm)
        if left == 0:
            linearize_pgr(e, group_mfmas(1), pend, [0, 0])
            return
        if first:
            e(f"s_waitcnt vmcnt({8 + vm[1]})")                   # k tile 1 (LDS-DMA) is in stage 1: only k tile 2's loads may be out
        e("s_waitcnt lgkmcnt(0)")
        if PABL != "nobar":
            e("s_barrier")
        if left >= 2:
            e("s_add_u32 s82, s82, 128")
        ops = slots(group_mfmas(1), frag_reads(s ^ 1, 0, 0), xl)
        pend = linearize_pgr(e, ops, [], [0, 0])
        if left >= 3:
            e("s_add_u32 s81, s81, 128")
        assert pend == LOOP_PENDING or PABL == "noread", (pend, LOOP_PENDING)

    def step_check(tail):
        e(f"s_sub_u32 s83, {S_NK}, s84")                        # k steps left, this one included (>= 3 here)
        e("s_cmp_eq_u32 s83, 3")
        e(f"s_cbranch_scc1 {tail}")

    TAIL1, TAIL0 = e.lab("tail1"), e.lab("tail0")
    body(0, first=True)                                         # k step 0 (K >= 4 k tiles: 3 steps follow)
    e("s_mov_b32 s84, 1")
    e.label(LOOP)
    step_check(TAIL1)
    body(1)
    e("s_add_u32 s84, s84, 1")
    step_check(TAIL0)
    body(0)
    e("s_add_u32 s84, s84, 1")
    e(f"s_branch {LOOP}")
    e.label(TAIL1)
    body(1, left=2)
    body(0, left=1)
    body(1, left=0)
    e(f"s_branch {DONE}")
    e.label(TAIL0)
    body(0, left=2)
    body(1, left=1)
    body(0, left=0)
    e.label(DONE)
    e("s_waitcnt vmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                              # every wave is done with the LDS stages
    for _ in range(3):
        e("s_nop 7")                                            # last MFMA results readable by VALU


# ---------------------------------------------------------------- epilogues
T = 32                       # v[32:47] tile values
GV, BV = 48, 64              # v[48:63] gate runs (GELU: temporaries), v[64:79] bias runs
CO = 80                      # v[80:127] ring of three prefetched C tiles (resid)
VSEL, VCOLT = 12, 13         # per-row-block gate vector address, column temporary
KC = 14                      # v[14:19] GELU constant pairs
GE = 20                      # v[20:27] GELU temporaries
NTILES = NI * NJ
VLW, VLR, VCOL, VROW = "v28", "v29", "v30", "v31"   # LDS write / read addresses, first column + 8 h, row in the wave's patch
VCL, VH = "v20", "v21"       # 4 lane, 32 h (column_vectors only)


def column_vectors(e, kind):
    """Per wave: the vectors of its 192 columns in a wave-private LDS region (free after the k loop's last barrier):
    +0 gate for rows below the batch boundary, +768 bias, +1536 gate for rows from the boundary on.  Lane l handles
    columns l, l + 64, l + 128; out-of-range columns / absent vectors read 0 through the descriptors."""
    e(f"v_lshlrev_b32 {VCL}, 2, %[vlane]")
    e(f"v_and_b32 {VH}, 32, %[vlane]")
    e(f"v_and_b32 {VROW}, 31, %[vlane]")
    e(f"v_add_u32 {VLW}, {S_REG}, {VCL}")                        # region + 4 lane
    e(f"v_add_u32 {VLR}, {S_REG}, {VH}")                         # region + 32 h
    e(f"s_lshr_b32 {S_COLN}, {S_COL0}, 2")
    e(f"v_lshrrev_b32 {VCOL}, 2, {VH}")
    e(f"v_add_u32 {VCOL}, {S_COLN}, {VCOL}")                     # first column of the wave + 8 h
    for t in range(3):
        e(f"buffer_load_dword v{48 + t}, {VCL}, %[rbias], {S_COL0} offen offset:{t * 256}")        # bias[n]
        if kind == "resid":
            e(f"buffer_load_dword v{52 + t}, {VCL}, %[rg0], {S_COL0} offen offset:{t * 256}")      # gate0[n]
            e(f"buffer_load_dword v{56 + t}, {VCL}, %[rg1], {S_G1LO} offen offset:{t * 256}")      # gate1[b][n]
    if kind == "resid":
        e(f"s_add_u32 s85, {S_G1LO}, {S_G1ST}")
        for t in range(3):
            e(f"buffer_load_dword v{60 + t}, {VCL}, %[rg1], s85 offen offset:{t * 256}")           # gate1[b + 1][n]
    # the workgroup's next tile: its prologue DMA goes out now (the stages are free) and lands under this epilogue
    NONE, JOIN = e.lab("nonext"), e.lab("join")
    e(f"s_cmp_eq_u32 {S_NEXT}, 0")
    e(f"s_cbranch_scc1 {NONE}")
    tile_prologue(e, S_NXB, S_NWB)
    e(f"s_waitcnt vmcnt({8 + NW + pro_pieces()})")                # the column vectors (older than the prologue's DMA pieces)
    e(f"s_branch {JOIN}")
    e.label(NONE)
    e("s_waitcnt vmcnt(0)")
    e.label(JOIN)
    for t in range(3):
        e(f"ds_write_b32 {VLW}, v{48 + t} offset:{768 + t * 256}")
        if kind == "resid":
            e(f"v_add_f32 v{52 + t}, {S_GC}, v{52 + t}")                       # gate_const + gate0 (gemm_bf16.hip order)
            e(f"v_add_f32 v{56 + t}, v{52 + t}, v{56 + t}")
            e(f"v_add_f32 v{60 + t}, v{52 + t}, v{60 + t}")
            e(f"ds_write_b32 {VLW}, v{56 + t} offset:{t * 256}")
            e(f"ds_write_b32 {VLW}, v{60 + t} offset:{1536 + t * 256}")
    e("s_waitcnt lgkmcnt(0)")


def gelu_pairs(e, base):
    """gelu_tanh (omh_common.h, same operation order => same bits) on v[base:base+7] in place, two elements per packed
    instruction: x rcp(1 + 2^t), t = (k x) fma(0.044715, x^2, 1).  v[14:15] = 0.044715, v[16:17] = 1, v[18:19] = k."""
    for p in range(4):
        x, t, u = base + 2 * p, GE + 2 * p, GV + 2 * p
        e(f"v_pk_mul_f32 {vr(t, 2)}, {vr(x, 2)}, {vr(x, 2)}")
        e(f"v_pk_mul_f32 {vr(u, 2)}, {vr(KC + 4, 2)}, {vr(x, 2)}")
        e(f"v_pk_fma_f32 {vr(t, 2)}, {vr(KC, 2)}, {vr(t, 2)}, {vr(KC + 2, 2)}")
        e(f"v_pk_mul_f32 {vr(t, 2)}, {vr(u, 2)}, {vr(t, 2)}")
    for p in range(4):
        t = GE + 2 * p
        e(f"v_exp_f32 v{t}, v{t}")
        e(f"v_exp_f32 v{t + 1}, v{t + 1}")
    for p in range(4):
        t = GE + 2 * p
        e(f"v_pk_add_f32 {vr(t, 2)}, {vr(KC + 2, 2)}, {vr(t, 2)}")
    for p in range(4):
        t = GE + 2 * p
        e(f"v_rcp_f32 v{t}, v{t}")
        e(f"v_rcp_f32 v{t + 1}, v{t + 1}")
    e("s_nop 0")
    for p in range(4):
        x, t = base + 2 * p, GE + 2 * p
        e(f"v_pk_mul_f32 {vr(x, 2)}, {vr(x, 2)}, {vr(t, 2)}")


def c_loads(e, n, es):
    """The four 16-byte pieces per lane of tile n's C values -> ring slot n % 3; s88 = soffset of tile n's row block."""
    i = n % NI
    for p in range(2):
        for q in range(2):
            e(f"buffer_load_dwordx4 {vr(CO + (n % 3) * 16 + p * 8 + q * 4, 4)}, %[voc], %[rc], s88 offen "
              f"offset:{(i * 32 + 16 * p) * es + q * 16}")


RBM = 120                    # bf16m: v[120:123] = bias of the lane's row in row blocks j = 0..3 (per-ROW bias, OMH_BIAS_M)


def epilogue(e, kind):
    # "geluaux": the gelu stream that also writes the pre-activation u_pre = bf16(acc + bias) to `aux` (same shape and row
    # pitch as C: a lane's aux bytes sit at its C store offset) — the FFN-up projection of a training forward whose block
    # is back-propagated (OMH_EPI_GELU_BF16 with aux, ABI v5; model.py:272-274 under autograd).  The converted words go
    # through a ring of four 4-register slots in v[80:95] (free in every kind but "resid").
    aux = kind == "geluaux"
    if aux:
        kind = "gelu"
    out_bf16 = kind in ("bf16", "gelu", "bf16m")
    rowbias = kind == "bf16m"    # C = bf16(acc + bias[m]): the V^T projection (operands swapped, model.py:152-153 via :214)
    resid = kind == "resid"
    es = 2 if out_bf16 else 4
    if rowbias:                  # older than the next tile's prologue DMA: column_vectors' wait covers them
        for j in range(NJ):
            e(f"buffer_load_dword v{RBM + j}, %[vbm], %[rbm], 0 offen offset:{j * 128}")
    column_vectors(e, "bf16" if rowbias else kind)
    if kind == "gelu":
        for r_, val in ((KC, "0x3d372713"), (KC + 1, "0x3d372713"), (KC + 2, "1.0"), (KC + 3, "1.0"),
                        (KC + 4, "0xc0135761"), (KC + 5, "0xc0135761")):
            e(f"v_mov_b32 v{r_}, {val}")
    # tiles in the order n = 6 j + i (row block j, column tile i)
    e(f"s_mov_b32 s85, {S_SCB}")                                 # store soffset: row block of the tile being written
    e(f"s_mov_b32 s88, {S_SCB}")                                 # load soffset: row block of the tile being fetched
    if resid:
        c_loads(e, 0, es)
        c_loads(e, 1, es)
    for n in range(NTILES):
        j, i = divmod(n, NI)
        t = i * NJ + j
        if i == 0:
            if j:
                e(f"s_add_u32 s85, s85, {S_SCJ}")
                e(f"v_add_u32 {VROW}, 32, {VROW}")
            if resid:
                e(f"v_cmp_le_u32 vcc, {S_MB}, {VROW}")
                e(f"v_add_u32 v{VCOLT}, 1536, {VLR}")
                e(f"v_cndmask_b32 v{VSEL}, {VLR}, v{VCOLT}, vcc")
        if resid and n + 2 < NTILES:
            if (n + 2) % NI == 0:
                e(f"s_add_u32 s88, s88, {S_SCJ}")
            c_loads(e, n + 2, es)
        for p in range(2):
            col = (i * 32 + 16 * p) * 4
            if resid:
                e(f"ds_read_b128 {vr(GV + p * 8, 4)}, v{VSEL} offset:{col}")
                e(f"ds_read_b128 {vr(GV + p * 8 + 4, 4)}, v{VSEL} offset:{col + 16}")
            if rowbias:
                for r_ in range(8):
                    e(f"v_mov_b32 v{BV + p * 8 + r_}, v{RBM + j}")
            else:
                e(f"ds_read_b128 {vr(BV + p * 8, 4)}, {VLR} offset:{768 + col}")
                e(f"ds_read_b128 {vr(BV + p * 8 + 4, 4)}, {VLR} offset:{768 + col + 16}")
        for r_ in range(16):
            if t < 16:
                e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
            else:
                e(f"v_mov_b32 v{T + r_}, v{128 + (t - 16) * 16 + r_}")
        e("s_nop 1")
        for q0 in (0, 2):                                        # quads (0,1), (2,3) -> two runs of 8 consecutive n
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        e("s_waitcnt lgkmcnt(0)")
        if resid:
            # in issue order behind tile n's loads: S(n-2) L(n+1) S(n-1) L(n+2), four instructions each
            behind = 4 * ((n >= 2) + (n + 1 < NTILES) + (n >= 1) + (n + 2 < NTILES))
            e(f"s_waitcnt vmcnt({behind})")
        for p in range(2):
            v0 = T + 8 * p
            e(f"v_add_u32 v{VCOLT}, {i * 32 + 16 * p}, {VCOL}")
            e(f"v_cmp_gt_u32 vcc, {S_N}, v{VCOLT}")
            e("s_and_saveexec_b64 s[86:87], vcc")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(BV + p * 8 + r_, 2)}")
            if resid:
                for r_ in range(0, 8, 2):
                    e(f"v_pk_mul_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(GV + p * 8 + r_, 2)}")
                for r_ in range(0, 8, 2):
                    e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(CO + (n % 3) * 16 + p * 8 + r_, 2)}, {vr(v0 + r_, 2)}")
            off = (i * 32 + 16 * p) * es
            if aux:
                ax = CO + ((2 * n + p) % 4) * 4
                for r_ in range(4):
                    e(f"v_cvt_pk_bf16_f32 v{ax + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
                e(f"buffer_store_dwordx4 {vr(ax, 4)}, %[voc], %[raux], s85 offen offset:{off}")
            if kind == "gelu":
                gelu_pairs(e, v0)
            if out_bf16:
                for r_ in range(4):
                    e(f"v_cvt_pk_bf16_f32 v{v0 + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
                e(f"buffer_store_dwordx4 {vr(v0, 4)}, %[voc], %[rc], s85 offen offset:{off}")
            else:
                e(f"buffer_store_dwordx4 {vr(v0, 4)}, %[voc], %[rc], s85 offen offset:{off}")
                e(f"buffer_store_dwordx4 {vr(v0 + 4, 4)}, %[voc], %[rc], s85 offen offset:{off + 16}")
            e("s_nop 1")
            e("s_mov_b64 exec, s[86:87]")


def epilogue_resid(e):
    """C fp32 += (acc + bias) * gate.  The old C values are what the epilogue waits for (786 KB per tile and CU): the
    more of them are in flight the better, and registers are the limit.  The 8 accumulator tiles that live in
    v[128:255] therefore go first — each frees 16 registers when it has been read — and from then on up to 11 tiles
    of old values are in flight instead of 3 (ring slots: v[80:127] + the freed accumulator registers)."""
    es = 4
    depth = int(os.environ.get("OMH_GW64_RING", "11"))
    # experiment knob, off: a second permlane level gives a lane 16 CONSECUTIVE columns (64 B per lane, full 128-B lines
    # per row and instruction pair) — measured 197 -> 202 us on the o-projection: the epilogue is not bound by request count
    W16 = os.environ.get("OMH_GW64_WIDE16", "0") == "1"
    column_vectors(e, "resid")
    VLR_, VCOL_, VOC_ = VLR, VCOL, "%[voc]"
    if W16:
        e(f"v_add_u32 v24, {VLR}, {VH}")                         # vector region + 64 h bytes
        e(f"v_add_u32 v25, {VH}, %[voc]")                        # row offset + 16 h columns
        e(f"v_lshrrev_b32 v26, 2, {VH}")
        e(f"v_add_u32 v26, v26, {VCOL}")                         # first column of the wave + 16 h
        VLR_, VCOL_, VOC_ = "v24", "v26", "v25"
    seq = [(4 + m // 4, m % 4) for m in range(8)] + [(t // NJ, t % NJ) for t in range(16)]
    free = [CO, CO + 16, CO + 32][:min(3, depth)]
    slot_of, issued, nxt = {}, [], [0]

    def request():
        """Old C values of the next tiles in processing order, while a ring slot is free."""
        while nxt[0] < NTILES and free:
            k = nxt[0]
            i, j = seq[k]
            sl = free.pop(0)
            slot_of[k] = sl
            e(f"s_mul_i32 s88, {S_SCJ}, {j}")
            e(f"s_add_u32 s88, s88, {S_SCB}")
            for p in range(2):
                for q in range(2):
                    off = (i * 32) * es + p * 32 + q * 16 if W16 else (i * 32 + 16 * p) * es + q * 16
                    e(f"buffer_load_dwordx4 {vr(sl + p * 8 + q * 4, 4)}, {VOC_}, %[rc], s88 offen offset:{off}")
                    issued.append(("L", k))
            nxt[0] += 1

    request()
    for k, (i, j) in enumerate(seq):
        t = i * NJ + j
        # row block j: store soffset, gate vector set of this lane's row
        e(f"s_mul_i32 s85, {S_SCJ}, {j}")
        e(f"s_add_u32 s85, s85, {S_SCB}")
        e(f"v_add_u32 v{VCOLT}, {32 * j}, {VROW}")
        e(f"v_cmp_le_u32 vcc, {S_MB}, v{VCOLT}")
        e(f"v_add_u32 v{VCOLT}, 1536, {VLR_}")
        e(f"v_cndmask_b32 v{VSEL}, {VLR_}, v{VCOLT}, vcc")
        for p in range(2):
            col = (i * 32) * 4 + p * 32 if W16 else (i * 32 + 16 * p) * 4
            e(f"ds_read_b128 {vr(GV + p * 8, 4)}, v{VSEL} offset:{col}")
            e(f"ds_read_b128 {vr(GV + p * 8 + 4, 4)}, v{VSEL} offset:{col + 16}")
            e(f"ds_read_b128 {vr(BV + p * 8, 4)}, {VLR_} offset:{768 + col}")
            e(f"ds_read_b128 {vr(BV + p * 8 + 4, 4)}, {VLR_} offset:{768 + col + 16}")
        for r_ in range(16):
            if t < 16:
                e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
            else:
                e(f"v_mov_b32 v{T + r_}, v{128 + (t - 16) * 16 + r_}")
        if t >= 16 and len(slot_of) + len(free) - k < depth:      # the accumulator tile's registers join the ring
            free.append(128 + (t - 16) * 16)
        e("s_nop 1")
        for q0 in (0, 2):                                        # quads (0,1), (2,3) -> two runs of 8 consecutive n
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        if W16:                                                  # ... and the two runs -> 16 consecutive n per lane
            e("s_nop 1")
            for r_ in range(8):
                e(f"v_permlane32_swap_b32 v{T + r_}, v{T + 8 + r_}")
        request()
        e("s_waitcnt lgkmcnt(0)")
        last = max(x for x, tag in enumerate(issued) if tag == ("L", k))
        e(f"s_waitcnt vmcnt({min(63, len(issued) - last - 1)})")     # in-order counter: tile k's old values have landed
        sl = slot_of[k]
        for p in range(2):
            v0 = T + 8 * p
            e(f"v_add_u32 v{VCOLT}, {i * 32 + (8 if W16 else 16) * p}, {VCOL_}")
            e(f"v_cmp_gt_u32 vcc, {S_N}, v{VCOLT}")
            e("s_and_saveexec_b64 s[86:87], vcc")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(BV + p * 8 + r_, 2)}")
            for r_ in range(0, 8, 2):
                e(f"v_pk_mul_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(GV + p * 8 + r_, 2)}")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(sl + p * 8 + r_, 2)}, {vr(v0 + r_, 2)}")
            off = (i * 32) * es + p * 32 if W16 else (i * 32 + 16 * p) * es
            e(f"buffer_store_dwordx4 {vr(v0, 4)}, {VOC_}, %[rc], s85 offen offset:{off}")
            e(f"buffer_store_dwordx4 {vr(v0 + 4, 4)}, {VOC_}, %[rc], s85 offen offset:{off + 16}")
            issued += [("S", k), ("S", k)]
            e("s_nop 1")
            e("s_mov_b64 exec, s[86:87]")
        free.append(sl)
    assert nxt[0] == NTILES


def epilogue_gelubwd(e):
    """C bf16 = (acc + bias) * gelu_tanh'(aux): the FFN dgrad GEMM of the training step (du_pre = (dy W2) gelu'(u_pre),
    model.py:272-274 under autograd; OMH_EPI_GELU_BWD_BF16).  aux = the bf16 pre-activations, same shape and row pitch as
    C, so a lane's aux bytes sit at its C store offset; they are requested two accumulator tiles ahead (ring of three
    8-register slots).  gelu_tanh_grad of omh_common.h operation by operation (same bits as the 8-wave kernel):
        s = rcp(1 + exp2((k x) fma(a, x^2, 1)));  g = fma((x 2c) fma(3a, x^2, 1), s (1 - s), s)."""
    es = 2
    column_vectors(e, "bf16")
    AR = 80                                                      # v[80:103]: ring of three tiles' aux words (2 x 4 each)
    X, X2, TT, SS = 104, 112, 120, 20                            # v[104:111] x, v[112:119] x^2, v[120:127] temporaries, v[20:27] s
    KA, K1, KK, K2C, K3A, KM1 = 48, 50, 52, 54, 56, 58           # constant pairs in v[48:59] (GV is free for this kind)
    for r_, val in ((KA, "0x3d372713"), (K1, "1.0"), (KK, "0xc0135761"), (K2C, "0x3fcc422a"), (K3A, "0x3e095d4e"),
                    (KM1, "-1.0")):
        e(f"v_mov_b32 v{r_}, {val}")
        e(f"v_mov_b32 v{r_ + 1}, {val}")

    def aux_loads(n):
        i = n % NI
        for p in range(2):
            e(f"buffer_load_dwordx4 {vr(AR + (n % 3) * 8 + p * 4, 4)}, %[voc], %[raux], s88 offen offset:{(i * 32 + 16 * p) * es}")

    e(f"s_mov_b32 s85, {S_SCB}")
    e(f"s_mov_b32 s88, {S_SCB}")
    aux_loads(0)
    aux_loads(1)
    for n in range(NTILES):
        j, i = divmod(n, NI)
        t = i * NJ + j
        if i == 0 and j:
            e(f"s_add_u32 s85, s85, {S_SCJ}")
            e(f"v_add_u32 {VROW}, 32, {VROW}")
        if n + 2 < NTILES:
            if (n + 2) % NI == 0:
                e(f"s_add_u32 s88, s88, {S_SCJ}")
            aux_loads(n + 2)
        for p in range(2):
            col = (i * 32 + 16 * p) * 4
            e(f"ds_read_b128 {vr(BV + p * 8, 4)}, {VLR} offset:{768 + col}")
            e(f"ds_read_b128 {vr(BV + p * 8 + 4, 4)}, {VLR} offset:{768 + col + 16}")
        for r_ in range(16):
            if t < 16:
                e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
            else:
                e(f"v_mov_b32 v{T + r_}, v{128 + (t - 16) * 16 + r_}")
        e("s_nop 1")
        for q0 in (0, 2):
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        e("s_waitcnt lgkmcnt(0)")
        # in issue order behind tile n's aux loads: S(n-2) L(n+1) S(n-1) L(n+2), two instructions each
        behind = 2 * ((n >= 2) + (n + 1 < NTILES) + (n >= 1) + (n + 2 < NTILES))
        e(f"s_waitcnt vmcnt({behind})")
        for p in range(2):
            v0 = T + 8 * p
            aw = AR + (n % 3) * 8 + p * 4
            e(f"v_add_u32 v{VCOLT}, {i * 32 + 16 * p}, {VCOL}")
            e(f"v_cmp_gt_u32 vcc, {S_N}, v{VCOLT}")
            e("s_and_saveexec_b64 s[86:87], vcc")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(BV + p * 8 + r_, 2)}")
            for w_ in range(4):                                  # bf16 pair -> two fp32
                e(f"v_lshlrev_b32 v{X + 2 * w_}, 16, v{aw + w_}")
                e(f"v_and_b32 v{X + 2 * w_ + 1}, 0xffff0000, v{aw + w_}")
            for q in range(4):
                x, x2, tt = X + 2 * q, X2 + 2 * q, TT + 2 * q
                e(f"v_pk_mul_f32 {vr(x2, 2)}, {vr(x, 2)}, {vr(x, 2)}")                       # x^2
                e(f"v_pk_mul_f32 {vr(tt, 2)}, {vr(KK, 2)}, {vr(x, 2)}")                      # k x
                e(f"v_pk_fma_f32 {vr(SS + 2 * q, 2)}, {vr(KA, 2)}, {vr(x2, 2)}, {vr(K1, 2)}")  # fma(a, x^2, 1)
                e(f"v_pk_mul_f32 {vr(tt, 2)}, {vr(tt, 2)}, {vr(SS + 2 * q, 2)}")             # (k x) * ...
            for q in range(8):
                e(f"v_exp_f32 v{TT + q}, v{TT + q}")
            for q in range(4):
                e(f"v_pk_add_f32 {vr(TT + 2 * q, 2)}, {vr(K1, 2)}, {vr(TT + 2 * q, 2)}")      # 1 + 2^t
            for q in range(8):
                e(f"v_rcp_f32 v{SS + q}, v{TT + q}")                                         # s
            e("s_nop 0")
            for q in range(4):
                x, x2, tt, ss = X + 2 * q, X2 + 2 * q, TT + 2 * q, SS + 2 * q
                e(f"v_pk_mul_f32 {vr(x, 2)}, {vr(x, 2)}, {vr(K2C, 2)}")                      # x * 2c
                e(f"v_pk_fma_f32 {vr(x2, 2)}, {vr(K3A, 2)}, {vr(x2, 2)}, {vr(K1, 2)}")       # fma(3a, x^2, 1)
                e(f"v_pk_mul_f32 {vr(x, 2)}, {vr(x, 2)}, {vr(x2, 2)}")                       # w
                e(f"v_pk_fma_f32 {vr(tt, 2)}, {vr(ss, 2)}, {vr(KM1, 2)}, {vr(K1, 2)}")       # 1 - s
                e(f"v_pk_mul_f32 {vr(tt, 2)}, {vr(ss, 2)}, {vr(tt, 2)}")                     # s (1 - s)
                e(f"v_pk_fma_f32 {vr(tt, 2)}, {vr(x, 2)}, {vr(tt, 2)}, {vr(ss, 2)}")         # g = fma(w, s(1-s), s)
                e(f"v_pk_mul_f32 {vr(v0 + 2 * q, 2)}, {vr(v0 + 2 * q, 2)}, {vr(tt, 2)}")     # v *= g
            off = (i * 32 + 16 * p) * es
            for r_ in range(4):
                e(f"v_cvt_pk_bf16_f32 v{v0 + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
            e(f"buffer_store_dwordx4 {vr(v0, 4)}, %[voc], %[rc], s85 offen offset:{off}")
            e("s_nop 1")
            e("s_mov_b64 exec, s[86:87]")


def epilogue_resid192(e):
    """C fp32 += (acc + bias) * gate on the 256 x 192 tile: the old C values are already in v[96:255] / a[192:223]
    (c_prefetch, requested during the k loop), so the epilogue is arithmetic and stores only — it neither waits for HBM
    reads nor holds the matrix pipe idle while ~64 lines per CU trickle in (the 256 x 384 stream's epilogue: 786 KB of
    read-modify-write per tile at ~10 B/clk/CU)."""
    es = 4
    column_vectors(e, "resid")
    # training epilogue (ABI v5): y = bf16(acc + bias) into `aux` — same row pitch as C, so every aux byte offset is half
    # the C offset; a null aux is an empty descriptor (the stores are dropped)
    VOA, AX = 24, 20                                             # v24: voc / 2; v[20:23]: packed bf16
    e(f"v_lshrrev_b32 v{VOA}, 1, %[voc]")
    for k in range(NTILES):
        i, j = divmod(k, NJ)
        t = i * NJ + j
        e(f"s_mul_i32 s85, {S_SCJ}, {j}")
        e(f"s_add_u32 s85, s85, {S_SCB}")
        e("s_lshr_b32 s88, s85, 1")
        e(f"v_add_u32 v{VCOLT}, {32 * j}, {VROW}")
        e(f"v_cmp_le_u32 vcc, {S_MB}, v{VCOLT}")
        e(f"v_add_u32 v{VCOLT}, 1536, {VLR}")
        e(f"v_cndmask_b32 v{VSEL}, {VLR}, v{VCOLT}, vcc")
        for p in range(2):
            col = (i * 32 + 16 * p) * 4
            e(f"ds_read_b128 {vr(GV + p * 8, 4)}, v{VSEL} offset:{col}")
            e(f"ds_read_b128 {vr(GV + p * 8 + 4, 4)}, v{VSEL} offset:{col + 16}")
            e(f"ds_read_b128 {vr(BV + p * 8, 4)}, {VLR} offset:{768 + col}")
            e(f"ds_read_b128 {vr(BV + p * 8 + 4, 4)}, {VLR} offset:{768 + col + 16}")
        for r_ in range(16):
            e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
        sl = CPRE + 16 * k
        if k >= 10:                                              # old C parked in AGPRs -> the slot tile k - 10 has left
            sl = CPRE + 16 * (k - 10)
            for r_ in range(16):
                e(f"v_accvgpr_read_b32 v{sl + r_}, a{192 + 16 * (k - 10) + r_}")
        e("s_nop 1")
        for q0 in (0, 2):                                        # quads (0,1), (2,3) -> two runs of 8 consecutive n
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        e("s_waitcnt lgkmcnt(0)")
        for p in range(2):
            v0 = T + 8 * p
            e(f"v_add_u32 v{VCOLT}, {i * 32 + 16 * p}, {VCOL}")
            e(f"v_cmp_gt_u32 vcc, {S_N}, v{VCOLT}")
            e("s_and_saveexec_b64 s[86:87], vcc")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(BV + p * 8 + r_, 2)}")
            off = (i * 32 + 16 * p) * es
            for r_ in range(4):
                e(f"v_cvt_pk_bf16_f32 v{AX + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
            e(f"buffer_store_dwordx4 {vr(AX, 4)}, v{VOA}, %[raux], s88 offen offset:{off // 2}")
            for r_ in range(0, 8, 2):
                e(f"v_pk_mul_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(GV + p * 8 + r_, 2)}")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(sl + p * 8 + r_, 2)}, {vr(v0 + r_, 2)}")
            e(f"buffer_store_dwordx4 {vr(v0, 4)}, %[voc], %[rc], s85 offen offset:{off}")
            e(f"buffer_store_dwordx4 {vr(v0 + 4, 4)}, %[voc], %[rc], s85 offen offset:{off + 16}")
            e("s_nop 1")
            e("s_mov_b64 exec, s[86:87]")


def epilogue_vt(e):
    """C^T bf16: out[n][m] = bf16(acc + bias[n]) — the V third of the fused q | k | v projection (model.py:144-146,
    152-153: V^T [d, S] is what the attention kernel reads).  The main loop ran with the MFMA operands swapped (SWAP), so
    after the permlane widening lane (r, h) holds, per accumulator tile (i, j) and run p, the 8 ROWS m = 32 j + 16 p +
    8 h .. of COLUMN n = 32 i + r of the wave's patch: 16 contiguous bytes of output row n.  S_SCB = the origin of the
    WAVE's patch in the output ((first n) * ld + first m) * 2, S_SCJ = 32 * ld * 2 (one i step; the lane adds r * ld * 2 + 16 h),
    S_MB = rows m left from the wave's first row (runs at or past it are masked: they would land in the pad columns),
    output rows past N fall outside the descriptor.  The bias is per lane: six registers (one per i)."""
    VBM, VOCT = "v15", "v16"
    e(f"v_and_b32 {VBM}, 31, %[vlane]")
    e(f"v_lshlrev_b32 {VBM}, 2, {VBM}")                          # 4 r: the lane's bias offset inside the wave's columns
    for i in range(NI):                # older than the next tile's prologue DMA: column_vectors' wait covers them
        e(f"buffer_load_dword v{RBM + i}, {VBM}, %[rbias], {S_COL0} offen offset:{i * 128}")
    column_vectors(e, "bf16")
    V8H, RB2 = "v14", 96                                         # v[96:107]: the lane's bias as a register PAIR per i
    e(f"v_lshrrev_b32 {V8H}, 2, {VH}")                           # 8 h
    # the lane's output offset: r rows of the output (pitch = S_SCJ / 32 bytes) + 8 h columns of 2 bytes
    e(f"s_lshr_b32 s88, {S_SCJ}, 5")
    e(f"v_and_b32 {VOCT}, 31, %[vlane]")
    e(f"v_mul_lo_u32 {VOCT}, {VOCT}, s88")
    e(f"v_lshl_add_u32 {VOCT}, {V8H}, 1, {VOCT}")
    for i in range(NI):
        e(f"v_mov_b32 v{RB2 + 2 * i}, v{RBM + i}")
        e(f"v_mov_b32 v{RB2 + 2 * i + 1}, v{RBM + i}")
    for n in range(NTILES):
        j, i = divmod(n, NI)
        t = i * NJ + j
        e(f"s_mul_i32 s85, {S_SCJ}, {i}")
        e(f"s_add_u32 s85, s85, {S_SCB}")
        for r_ in range(16):
            if t < 16:
                e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
            else:
                e(f"v_mov_b32 v{T + r_}, v{128 + (t - 16) * 16 + r_}")
        e("s_nop 1")
        for q0 in (0, 2):                                        # quads (0,1), (2,3) -> two runs of 8 consecutive m
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        for p in range(2):
            v0 = T + 8 * p
            e(f"v_add_u32 v{VCOLT}, {32 * j + 16 * p}, {V8H}")
            e(f"v_cmp_gt_u32 vcc, {S_MB}, v{VCOLT}")
            e("s_and_saveexec_b64 s[86:87], vcc")
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(RB2 + 2 * i, 2)}")
            for r_ in range(4):
                e(f"v_cvt_pk_bf16_f32 v{v0 + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
            e(f"buffer_store_dwordx4 {vr(v0, 4)}, {VOCT}, %[rc], s85 offen offset:{(32 * j + 16 * p) * 2}")
            e("s_nop 1")
            e("s_mov_b64 exec, s[86:87]")


# VMEM instructions PER ACCUMULATOR TILE an epilogue issues after the next tile's prologue DMA (the k loop's first wait
# counts them: an over-estimate would let k tile 0 be read before it has landed)
EPI_VMEM_TILE = {"f32": 4, "bf16": 2, "gelu": 2, "geluaux": 4, "resid": 8, "resid192": 6, "gelubwd": 4, "bf16m": 2, "bf16vt": 2}


def generate(kind, tag=None):
    global SWAP
    e = Emit(tag or kind)
    SWAP = kind == "bf16vt"
    if PGR:
        main_loop_pgr(e, EPI_VMEM_TILE[kind] * NTILES)
    else:
        main_loop(e, EPI_VMEM_TILE[kind] * NTILES, cpre=(kind == "resid192"))
    SWAP = False
    if kind == "bf16vt":
        epilogue_vt(e)
    elif kind == "resid" and NI == 6:
        epilogue_resid(e)
    elif kind == "resid192":
        epilogue_resid192(e)
    elif kind == "gelubwd":
        epilogue_gelubwd(e)
    else:
        epilogue(e, kind)
    return e


def first_prologue(tag="pro"):
    e = Emit(tag)
    unpack(e, (0, 2, 8))
    tile_prologue(e, S_NXB, S_NWB)
    e(f"s_waitcnt vmcnt({pro_pieces()})")
    return e


def main():
    print("// GENERATED by gen_gemm_w64.py — do not edit; edit the generator.")
    streams = [("PRO", first_prologue())] + [(kind.upper(), generate(kind)) for kind in KINDS + ("gelubwd", "bf16m", "geluaux", "bf16vt")]
    configure(3)                                                 # the 256 x 192 gated-residual stream (old C prefetched)
    streams += [("PRO192", first_prologue("pro192")), ("RESID192", generate("resid192")),
                ("F32_192", generate("f32", "f32n3")), ("BF16_192", generate("bf16", "bf16n3"))]
    # Round 6 experiment, NOT part of the shipped library (the tracked .inc is generated without it): the 256 x 256 streams
    # with two k tiles in flight in registers (main_loop_pgr).  OMH_GW64_P256=1 generates them (gemm_w64.hip compiles its
    # K_*_P kinds when the macros exist), OMH_GW64_ABLATIONS=1 adds five timing-only builds of the fp32 one
    # (OMH_GW64_ABL_SET=ldmod: cache-policy variants of its loads instead).  Measured (profiles/r06_gemm_p256.txt): bit for
    # bit the shipped kernels' results, 4-12 % SLOWER than the 256 x 384 LDS-DMA streams — the L2 -> CU fill path delivers
    # ~21 B/clk/CU however early the loads are issued, so bytes per flop (tile area) decide, not prefetch depth.
    if os.environ.get("OMH_GW64_P256", "0") == "1" or os.environ.get("OMH_GW64_ABLATIONS", "0") == "1":
        configure(4, pgr=True)
        streams += [("PRO256", first_prologue("pro256"))] + [(f"{kind.upper()}_P256", generate(kind, f"{kind}p4"))
                                                             for kind in ("f32", "bf16", "gelu", "resid")]
    if os.environ.get("OMH_GW64_ABLATIONS", "0") == "1":          # timing-only builds
        global PABL, LDMOD, M16
        if os.environ.get("OMH_GW64_ABL_SET", "abl") == "m16":    # the SHIPPED 256 x 384 fp32 stream on 16x16x32 MFMAs (garbage results)
            configure(6)
            M16 = True
            streams.append(("F32_M16", generate("f32", "f32m16")))
            M16 = False
            configure(4, pgr=True)
        if os.environ.get("OMH_GW64_ABL_SET", "abl") == "ldmod":
            for tag, mod in (("A", " nt"), ("B", " sc1"), ("C", " sc0"), ("D", " sc0 sc1"), ("E", " sc1 nt")):
                LDMOD = mod
                streams.append((f"F32_P256_{tag}", generate("f32", f"f32p4{tag.lower()}")))
            LDMOD = ""
        else:
            for tag, abl in (("A", "noload"), ("B", "nowrite"), ("C", "nomem"), ("D", "nobar"), ("E", "noread")):
                PABL = abl
                streams.append((f"F32_P256_{tag}", generate("f32", f"f32p4{tag.lower()}")))
            PABL = ""
    configure(6)
    for name, e in streams:
        print(f"#define OMH_GEMM_W64_ASM_{name} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        print(f"// {name}: {len(e.lines)} lines, {sum('v_mfma' in ln for ln in e.lines)} MFMA")
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(80, 92)] + [f'"v{i}"' for i in range(12, 256)] + \
           [f'"a{i}"' for i in range(256)]
    print("#define OMH_GEMM_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))


if __name__ == "__main__":
    main()
