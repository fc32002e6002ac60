#!/usr/bin/env python3
"""Generator of the instruction streams of ``conv_cl_w64_kernel`` (conv_w64.hip): the kw-shared 3x3(x3) "same"
convolution of the VAE's residual blocks (vae.py:17-36 under :186-220) as ONE wave per SIMD — 4 waves, each a
128(voxels) x 96(couts) patch = 4 x 3 MFMA tiles, the 192 fp32 accumulators in AGPRs — on the same staging scheme as
``conv_cl_kw3_kernel`` (vae_conv.hip: a stage = one (kt, kh) tap pair x one 32-channel block for all three kw taps;
A slab [WBM voxels][32 ch] fetched once, B tile [WBN couts][3 x 32]), with THREE stages in LDS.

Why: the 8-wave kernel has 36 MFMAs per wave between two barriers and waits for a stage's DMA one stage after issuing
it; it runs at 0.30 of the MFMA peak although its bytes per flop allow > 0.8.  Here a stage is 72 MFMAs per wave,
its 13 LDS-DMA pieces per wave are issued TWO stages ahead, spread between the MFMAs, and the fragment reads carry
exact lgkmcnt waits (gen_gemm_w64.py has the same design for the GEMM).

    python gen_conv_w64.py > conv_w64_asm.inc

Two tile configurations share the per-wave code: P = 512 x 96 (4 waves stacked along the voxels; NA = 8 A pieces and
NB = 5 B pieces per wave and stage), Q = 256 x 192 (2 x 2 waves; NA = 4, NB = 9).  Stage layout (53 312 bytes):
[64 pad | A slab | B tile | 2 KiB sink for the disabled pieces]; output row j of a strip reads slab rows j-1, j, j+1
(the pad is row -1 of the slab), rows 0 and WBM-1 of a tile are computed and dropped (tiles advance by WBM-2 voxels),
the x neighbours of an image row's end voxels are zeroed with per-lane AND words.  Per-lane constants come from a
table the C++ prologue writes to LDS (44 dwords per lane: inline asm takes 30 operands).

Accumulation order per output element = conv_cl_kw3_kernel's (stages in (kt, kh, channel block) order, groups kw-major,
one v_mfma_f32_32x32x16_bf16 per 16 channels), the epilogue adds bias then residual as wide_epilogue does: the two
kernels agree bit for bit.

Register map: a[0:191] accumulators (tile = 4 i + j, i = cout tile, j = voxel tile); v[12:55] the lane table
(a_off2[8] a_y[8] w_off[9] xaddr[kw][half] waddr[half] edge0[4] edge2[4] voc rowmask); v[56:83] / v[84:111] fragment
buffers (3 W + 4 X quads); v[112:119] DMA temporaries; epilogue: v[56:71] tile values, v[120:167] bias runs,
v[168:215] ring of three residual tiles.  s[60:73] scalar arguments, s[80:99] state and scratch.
"""
import os
import sys

NI, NJ = 3, 4
ABL = os.environ.get("OMH_CW64_ABL", "")     # timing-only ablations (wrong results): "dma", "and", "bar", "reads"
STAGE = 53312
NSTAGES = 3
KINDS = ("bf16", "f32")
CONFIGS = {"P": (8, 5), "Q": (4, 9)}

# scalar arguments (pairs bound to s[60:73] by conv_w64.hip)
S_LDSA0, S_LDSB0, S_ROWB, S_HIN, S_CBLK, S_NS, S_FRAME = "s60", "s61", "s62", "s63", "s64", "s65", "s66"
S_CWRAPA, S_CWRAPW, S_BLAST, S_JSTEP, S_SPARE0, S_SPARE1, S_SPARE2 = "s67", "s68", "s69", "s70", "s71", "s72", "s73"
S_UP = S_SPARE0                                   # 1: the input is read through a folded nearest-2x upsample (vae.py:76-79), else 0
# state: s80 LDS offset of the buffer being FETCHED, s81 / s82 source offsets of the A / B stage being fetched,
# s83 channel block, s84 kh, s85 loop counter, s[86:87] exec save, s89 compute buffer
# index, s90 / s91 piece bases, s92..s95 temporaries, s97 / s98 stage step constants
A_OFF2, A_Y, W_OFF, XADDR, WADDR, EDGE0, EDGE2, VOC, ROWMASK = 12, 20, 28, 37, 43, 45, 49, 53, 54
FRAG = (56, 84)
TMP = 112
T = 56
BIASV = 72                                        # epilogue: the bias runs take over the dead fragment buffers
V_RES_OFF, V_ST_OFF, V_ROWBIT, V_LANE, V_PAIR = 248, 249, 250, 251, 252    # v[248:253]: free during the k loop as well
# The residual tile (the fp32 trunk a block's second convolution adds: 48 KiB per wave, 192 KiB per CU) is requested
# DURING the k loop, one 32 x 32 tile per stage from the first 12 stages, into registers the loop does not use:
# v[120:247] (8 tiles) and a[192:255] (4 tiles; VMEM loads write AGPRs directly).  Round 2 requested it in the epilogue
# through a ring seven tiles deep: with every request an HBM miss, the CU's ~64 lines in flight bound the read to
# ~10 B/clk and the matrix pipe sat idle for 15 us per tile (0.32 vs 0.39 of the MFMA peak, fp32-trunk kinds vs bf16).
UNROLL = 13                                       # stages peeled at the head of the loop (12 carry a tile's requests)


def res_slot(n, kind):
    if kind == "bf16":
        return "v", 120 + 8 * n
    return ("v", 120 + 16 * n) if n < 8 else ("a", 192 + 16 * (n - 8))


def reg(bank, lo, n=1):
    return f"{bank}{lo}" if n == 1 else f"{bank}[{lo}:{lo + n - 1}]"


def res_request(n, kind):
    """Instructions that request residual tile n = 3 j + i (v248 = the lane's byte offset in row block j: the row
    block goes in the VGPR, not in the soffset, because the range check sees only voffset + inst_offset and tile 0's
    lane of row -1 must be out of range for j = 0 ONLY)."""
    i = n % NI
    es = 2 if kind == "bf16" else 4
    bank, base = res_slot(n, kind)
    out = []
    if n and i == 0:
        out.append(f"v_add_u32 v{V_RES_OFF}, {S_JSTEP}, v{V_RES_OFF}")
    for p in range(2):
        if kind == "bf16":
            out.append(f"buffer_load_dwordx4 {reg(bank, base + p * 4, 4)}, v{V_RES_OFF}, %[rres], 0 offen offset:{(i * 32 + 16 * p) * es}")
        else:
            for q in range(2):
                out.append(f"buffer_load_dwordx4 {reg(bank, base + p * 8 + q * 4, 4)}, v{V_RES_OFF}, %[rres], 0 offen "
                           f"offset:{(i * 32 + 16 * p) * es + q * 16}")
    return out


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def acc(i, j):
    t = i * NJ + j
    return f"a[{t * 16}:{t * 16 + 15}]"


def wfrag(buf, i):
    return FRAG[buf] + 4 * i


def xfrag(buf, j):
    return FRAG[buf] + 12 + 4 * j


class Emit:
    def __init__(self, tag):
        self.lines, self.tag = [], tag

    def __call__(self, s):
        self.lines.append(s)

    def lab(self, name):
        return f".Lcw64{self.tag}_{name}_%="

    def label(self, name):
        self.lines.append(f"{name}:")

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


def linearize(e, ops, pending):
    """ops: ("r", tag, text) LDS read; ("m", text, [tags]) instruction that needs those reads landed; ("x", text)."""
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


def frag_reads(G, buf):
    """The 3 W + 4 X fragment reads of group G (tap kw = G >> 1, 16-channel half G & 1) from the buffer the address
    registers point at."""
    kw, half = G >> 1, G & 1
    out = []
    for i in range(NI):
        out.append(("r", f"W{buf}.{i}", f"ds_read_b128 {vr(wfrag(buf, i), 4)}, {vr(WADDR + half)} offset:{kw * 64 + i * 6144}"))
    for j in range(NJ):
        out.append(("r", f"X{buf}.{j}", f"ds_read_b128 {vr(xfrag(buf, j), 4)}, {vr(XADDR + 2 * kw + half)} offset:{j * 2048}"))
    return out


def group_ops(G, buf, first=False):
    """MFMAs of group G, voxel tile major; the dx = -1 / +1 taps first AND the voxel fragment with the lane's edge
    word (an image row's end voxels have no such neighbour)."""
    kw = G >> 1
    out = []
    for j in range(NJ):
        if kw != 1 and ABL != "and":
            m = (EDGE0 if kw == 0 else EDGE2) + j
            for r_ in range(4):
                out.append(("m", f"v_and_b32 v{xfrag(buf, j) + r_}, v{m}, v{xfrag(buf, j) + r_}", [f"X{buf}.{j}"]))
            out.append(("x", "s_nop 1"))
        for i in range(NI):
            c = "0" if first else acc(i, j)
            out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(wfrag(buf, i), 4)}, {vr(xfrag(buf, j), 4)}, {c}",
                        [f"W{buf}.{i}", f"X{buf}.{j}"]))
    return out


def and_ops(G, buf, j):
    """The four ANDs of voxel fragment j of group G with the lane's edge word ([] for the centre tap)."""
    kw = G >> 1
    if kw == 1 or ABL == "and":
        return []
    m = (EDGE0 if kw == 0 else EDGE2) + j
    return [("m", f"v_and_b32 v{xfrag(buf, j) + r_}, v{m}, v{xfrag(buf, j) + r_}", [f"X{buf}.{j}"]) for r_ in range(4)]


def group_sched(G, buf, first, reads_next, dma_instrs, ands_next):
    """Group G as 12 MFMA slots (voxel tile major) with everything else dealt into the gaps behind them, a few
    instructions per gap (one wave per SIMD: whatever stalls the wave's issue for longer than an MFMA's 32 cycles is a
    bubble in the matrix pipe): the next group's 7 fragment reads in gaps 0..6, this group's edge ANDs of fragment
    j >= 1 in the gaps of fragment j - 1, the next group's ANDs of fragment 0 in gaps 9 / 10, the DMA instructions
    spread over all gaps."""
    gaps = [[] for _ in range(NJ * NI)]
    for k, r in enumerate(reads_next):
        gaps[k].append(r)
    for j in range(1, NJ):
        a = and_ops(G, buf, j)
        gaps[NI * (j - 1)] += a[:2]
        gaps[NI * (j - 1) + 1] += a[2:]
    gaps[9] += ands_next[:2]
    gaps[10] += ands_next[2:]
    n = len(dma_instrs)
    for t, ins in enumerate(dma_instrs):
        gaps[min(NJ * NI - 1, (t * NJ * NI) // n)].append(ins)
    out = []
    for j in range(NJ):
        for i in range(NI):
            c = "0" if first else acc(i, j)
            out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(wfrag(buf, i), 4)}, {vr(xfrag(buf, j), 4)}, {c}",
                        [f"W{buf}.{i}", f"X{buf}.{j}"]))
            out += gaps[NI * j + i]
    return out


def dma_pieces(NA, NB):
    """The 13 LDS-DMA pieces of the stage (s83 channel block, s84 kh, offsets s81 / s82) into the buffer at s80; then
    the scalar state moves on to the next stage.  Returns a list of op lists."""
    out = [[("x", f"s_add_u32 s90, s80, {S_LDSA0}"), ("x", f"s_add_u32 s91, s80, {S_LDSB0}")]]
    for q in range(NA):
        t = TMP + (q & 3)
        out.append([("x", f"v_add_u32 v{t}, s84, v{A_Y + q}"),                       # input row of the tap
                    ("x", f"v_cmp_gt_u32 vcc, {S_HIN}, v{t}"),                      # inside the (upsampled) image (unsigned)
                    ("x", f"v_lshrrev_b32 v{t}, {S_UP}, v{t}"),                     # folded nearest-2x upsample: input row y >> 1
                    ("x", f"v_mad_u32_u24 v{t}, v{t}, {S_ROWB}, v{A_OFF2 + q}"),
                    ("x", f"v_cndmask_b32 v{t}, v{TMP + 4}, v{t}, vcc"),            # outside: beyond the descriptor -> zeros
                    ("x", f"s_add_u32 m0, s90, {q * 1024}"),
                    ("x", f"buffer_load_dwordx4 v{t}, %[rx], s81 offen lds")])
    for q in range(NB):
        m0 = f"s_add_u32 m0, s80, {S_BLAST}" if q == NB - 1 else f"s_add_u32 m0, s91, {q * 1024}"
        out.append([("x", m0), ("x", f"buffer_load_dwordx4 v{W_OFF + q}, %[rw], s82 offen lds")])
    out.append([("x", "s_add_u32 s83, s83, 1"),
                ("x", f"s_cmp_eq_u32 s83, {S_CBLK}"),                               # last channel block of the tap pair
                ("x", "s_cselect_b32 s83, 0, s83"),
                ("x", f"s_cselect_b32 s92, {S_CWRAPA}, 64"),
                ("x", f"s_cselect_b32 s93, {S_CWRAPW}, 64"),
                ("x", "s_cselect_b32 s94, 1, 0"),
                ("x", "s_add_u32 s81, s81, s92"),
                ("x", "s_add_u32 s82, s82, s93"),
                ("x", "s_add_u32 s84, s84, s94"),
                ("x", "s_cmp_eq_u32 s84, 3"),                                       # next frame tap
                ("x", "s_cselect_b32 s84, 0, s84"),
                ("x", f"s_cselect_b32 s92, {S_FRAME}, 0"),
                ("x", "s_add_u32 s81, s81, s92"),
                ("x", f"s_add_u32 s80, s80, {STAGE}"),
                ("x", f"s_cmp_eq_u32 s80, {NSTAGES * STAGE}"),
                ("x", "s_cselect_b32 s80, 0, s80")])
    return out


def spread(mops, extras, lo=0, hi=None):
    """Insert the op lists `extras` evenly after the MFMAs (not the ANDs) of `mops`, MFMA index in [lo, hi)."""
    idx = [k for k, op in enumerate(mops) if op[0] == "m" and "v_mfma" in op[1]]
    hi = len(idx) if hi is None else hi
    n = hi - lo
    slots = {}
    for k, ex in enumerate(extras):
        pos = idx[lo + min(n - 1, (k * n) // max(1, len(extras)))]
        slots.setdefault(pos, []).extend(ex)
    out = []
    for k, op in enumerate(mops):
        out.append(op)
        out.extend(slots.get(k, []))
    return out


def advance(e):
    """Fragment addresses to the next buffer (s89 = index of the buffer they point at)."""
    e("s_add_u32 s89, s89, 1")
    e(f"s_cmp_eq_u32 s89, {NSTAGES}")
    e("s_cselect_b32 s89, 0, s89")
    e("s_cselect_b32 s92, s98, s97")                             # back to buffer 0 / one stage up
    for r_ in range(XADDR, XADDR + 8):
        e(f"v_add_u32 v{r_}, s92, v{r_}")


def main_loop(e, NA, NB, kind):
    NP = NA + NB
    R = 2 if kind == "bf16" else 4                               # VMEM instructions of one residual tile
    # the lane table, written to LDS by the C++ prologue
    for k in range(11):
        e(f"ds_read_b128 {vr(12 + 4 * k, 4)}, %[vtab] offset:{16 * k}")
    e(f"v_mov_b32 v{TMP + 4}, 0x80000000")                        # an offset outside every descriptor
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                               # the stages overwrite the table
    e(f"v_mov_b32 v{V_RES_OFF}, v{VOC}")                          # residual requests: the lane's row in row block 0
    for s_, v_ in (("s80", 0), ("s81", 0), ("s82", 0), ("s83", 0), ("s84", 0), ("s89", 0), ("s97", STAGE),
                   ("s98", (-(NSTAGES - 1) * STAGE) & 0xffffffff)):
        e(f"s_mov_b32 {s_}, {v_}")
    pieces = dma_pieces(NA, NB)
    for _ in range(2):                                           # stages 0 and 1
        for ops in pieces:
            for op in ops:
                e(op[1])
    e(f"s_waitcnt vmcnt({NP})")                                  # stage 0 has landed
    e("s_barrier")
    LOOP_PENDING = linearize(e, frag_reads(0, 0) + and_ops(0, 0, 0), [])
    e(f"s_sub_u32 s85, {S_NS}, 2")                               # steps that fetch a stage two ahead (>= UNROLL)

    def body(mode, first=False, res_tile=None, r_prev=0):
        """One stage = 6 groups of 12 MFMAs.  Groups 0..4: MFMAs || reads of the next group || (mode "full") the 13
        DMA pieces of the stage two ahead.  Then: the NEXT stage has landed (counted vmcnt: the in-order counter may
        leave the r_prev residual requests of the previous stage's tail and this stage's pieces outstanding),
        barrier (every wave has read this stage), addresses advance, group 5 || reads of the next stage's group 0 ||
        the requests of residual tile res_tile.  mode "last": the final stage, nothing to wait for or read ahead."""
        pend = LOOP_PENDING
        dm = pieces if (mode == "full" and ABL != "dma") else []
        per = [dm[0:4], dm[4:7], dm[7:10], dm[10:13], dm[13:]] if dm else [[]] * 5
        for G in range(5):
            reads = frag_reads(G + 1, (G + 1) & 1) if ABL != "reads" else []
            flat = [op for piece in per[G] for op in piece]
            ops = group_sched(G, G & 1, first and G == 0, reads, flat, and_ops(G + 1, (G + 1) & 1, 0))
            pend = linearize(e, ops, pend)
        if mode == "last":
            linearize(e, group_sched(5, 1, False, [], [], []), pend)
            return
        e(f"s_waitcnt vmcnt({(NP if (mode == 'full' and ABL != 'dma') else 0) + r_prev})")
        e("s_waitcnt lgkmcnt(0)")
        if ABL != "bar":
            e("s_barrier")
        advance(e)
        req = [("x", ln) for ln in res_request(res_tile, kind)] if res_tile is not None else []
        pend = linearize(e, group_sched(5, 1, False, frag_reads(0, 0), req, and_ops(0, 0, 0)), [])
        assert pend == LOOP_PENDING, (pend, LOOP_PENDING)

    LOOP = e.lab("loop")
    REST = e.lab("rest")
    for k in range(UNROLL):                                      # the head of the loop, peeled: tile k's residual
        body("full", first=(k == 0), res_tile=(k if k < NTILES else None), r_prev=(R if 0 < k <= NTILES else 0))
    e(f"s_sub_u32 s85, s85, {UNROLL}")
    e("s_cmp_eq_u32 s85, 0")
    e(f"s_cbranch_scc1 {REST}")
    e.label(LOOP)
    body("full")
    e("s_sub_u32 s85, s85, 1")
    e("s_cmp_lg_u32 s85, 0")
    e(f"s_cbranch_scc1 {LOOP}")
    e.label(REST)
    body("nodma")
    body("last")
    for _ in range(3):
        e("s_nop 7")                                             # last MFMA results readable by VALU


# ---------------------------------------------------------------- the split-bf16 ("pair") k loop, round 5
# The fp32-faithful VAE mode (vae.py:619-624: the reference's VAE computes in fp32) carries every operand as a bf16 pair
# hi = bf16(x), lo = bf16(x - hi) and needs x_hi w_hi + x_lo w_hi + x_hi w_lo per channel.  Rounds 3-4 ran that as an
# ordinary convolution over three channel blocks ([hi | lo | hi] against [hi | hi | lo]): x_hi and w_hi staged, read from
# LDS and masked TWICE.  Here the operands are interleaved per 16 channels — a slab row of 64 bytes is [x_hi(16) | x_lo(16)],
# a weight row per tap [w_hi(16) | w_lo(16)], i.e. an ordinary tensor of 2 C channels to the loader — and a stage's
# 3 taps x 3 products = 9 groups of 12 MFMAs share their fragments:
#     (kw, 0)  W_hi . X_hi   || reads X_lo(kw)
#     (kw, 1)  W_hi . X_lo   || reads W_lo(kw)
#     (kw, 2)  W_lo . X_hi   || reads W_hi(kw + 1), X_hi(kw + 1)
# 14 fragment reads, 13 LDS-DMA pieces, one barrier and at most 32 edge ANDs per 108 MFMAs where the three-block form
# has 21 / 19.5 / 1.5 / 48 per 108 — a third less of everything that is not an MFMA.  Register sets: W_A always holds
# w_hi, W_B w_lo; the two X sets swap roles with every tap (x_hi(kw + 1) goes where x_lo(kw) was, free after group
# (kw, 1)), so with 3 taps per stage the stream alternates between two stage bodies (parity 0 / 1) and needs an EVEN number
# of stages (9 taps x C / 16 blocks: always), 14 of them peeled at the head.  Accumulation order per output element:
# stages in (kt, kh, block) order, per block kw-major and hi.hi, lo.hi, hi.lo — fp32 sums in another order than the
# three-block form (1e-6 class, tests/test_gpu_vae.py).
UNROLL_PAIR = 14


def main_loop_pair(e, NA, NB, kind):
    NP = NA + NB
    R = 2 if kind == "bf16" else 4
    WSET = {"A": FRAG[0], "B": FRAG[1]}
    XSET = {0: FRAG[0] + 12, 1: FRAG[1] + 12}

    def wreads(h, kw, ws):
        return [("r", f"W{ws}.{i}", f"ds_read_b128 {vr(WSET[ws] + 4 * i, 4)}, {vr(WADDR + h)} offset:{kw * 64 + i * 6144}")
                for i in range(NI)]

    def xreads(h, kw, xs):
        return [("r", f"X{xs}.{j}", f"ds_read_b128 {vr(XSET[xs] + 4 * j, 4)}, {vr(XADDR + 2 * kw + h)} offset:{j * 2048}")
                for j in range(NJ)]

    def xands(kw, xs, j):
        if kw == 1:
            return []
        m = (EDGE0 if kw == 0 else EDGE2) + j
        return [("m", f"v_and_b32 v{XSET[xs] + 4 * j + r_}, v{m}, v{XSET[xs] + 4 * j + r_}", [f"X{xs}.{j}"]) for r_ in range(4)]

    def sched(ws, xs, first, reads_next, dma_instrs, own_kw, ands_next, read_gap0=0):
        """12 MFMAs W[ws] . X[xs] (voxel tile major) with the rest dealt into the gaps: the next group's fragment reads
        from gap read_gap0 on, the edge ANDs of this group's fragments j >= 1 (own_kw: the tap whose X set is fresh in this
        group, None when it was masked by an earlier group) in the gaps of fragment j - 1, the next group's ANDs of
        fragment 0 in gaps 9 / 10, the DMA instructions spread over all gaps."""
        gaps = [[] for _ in range(NJ * NI)]
        for k, r in enumerate(reads_next):
            gaps[min(NJ * NI - 1, read_gap0 + k)].append(r)
        if own_kw is not None:
            for j in range(1, NJ):
                a = xands(own_kw, xs, j)
                gaps[NI * (j - 1)] += a[:2]
                gaps[NI * (j - 1) + 1] += a[2:]
        gaps[9] += ands_next[:2]
        gaps[10] += ands_next[2:]
        n = len(dma_instrs)
        for t, ins in enumerate(dma_instrs):
            gaps[min(NJ * NI - 1, (t * NJ * NI) // n)].append(ins)
        out = []
        for j in range(NJ):
            for i in range(NI):
                c = "0" if first else acc(i, j)
                out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(WSET[ws] + 4 * i, 4)}, {vr(XSET[xs] + 4 * j, 4)}, {c}",
                            [f"W{ws}.{i}", f"X{xs}.{j}"]))
                out += gaps[NI * j + i]
        return out

    # the lane table, written to LDS by the C++ prologue (as main_loop)
    for k in range(11):
        e(f"ds_read_b128 {vr(12 + 4 * k, 4)}, %[vtab] offset:{16 * k}")
    e(f"v_mov_b32 v{TMP + 4}, 0x80000000")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e(f"v_mov_b32 v{V_RES_OFF}, v{VOC}")
    for s_, v_ in (("s80", 0), ("s81", 0), ("s82", 0), ("s83", 0), ("s84", 0), ("s89", 0), ("s97", STAGE),
                   ("s98", (-(NSTAGES - 1) * STAGE) & 0xffffffff)):
        e(f"s_mov_b32 {s_}, {v_}")
    pieces = dma_pieces(NA, NB)
    for _ in range(2):
        for ops in pieces:
            for op in ops:
                e(op[1])
    e(f"s_waitcnt vmcnt({NP})")
    e("s_barrier")
    # group (0, 0) of a stage of parity p needs W_hi(0) in W_A and X_hi(0) in X set p
    first_ops = lambda p: wreads(0, 0, "A") + xreads(0, 0, p) + xands(0, p, 0)
    LP = {0: linearize(e, first_ops(0), [])}
    LP[1] = [t.replace("X0.", "X1.") for t in LP[0]]
    e(f"s_sub_u32 s85, {S_NS}, 2")
    # the DMA op lists of a stage over groups 0 .. 7 (group 8 sits behind the stage's barrier)
    cut = [0, 2, 4, 6, 8, 10, 12, 14, len(pieces)]

    def body(p, mode, first=False, res_tile=None, r_prev=0):
        hi = lambda kw: (p + kw) & 1
        pend = LP[p]
        dm = pieces if mode == "full" else []
        per = [[op for piece in dm[cut[g]:cut[g + 1]] for op in piece] for g in range(8)] if dm else [[]] * 8
        for kw in range(3):
            h, l = hi(kw), 1 - hi(kw)
            # (kw, 0): W_hi . X_hi || X_lo(kw) -> the other X set (it held X_hi(kw - 1): its last MFMA was one group ago)
            ops = sched("A", h, first and kw == 0, xreads(1, kw, l), per[3 * kw], kw, xands(kw, l, 0), read_gap0=2)
            pend = linearize(e, ops, pend)
            # (kw, 1): W_hi . X_lo || W_lo(kw) -> W_B
            ops = sched("A", l, False, wreads(1, kw, "B"), per[3 * kw + 1], kw, [])
            pend = linearize(e, ops, pend)
            # (kw, 2): W_lo . X_hi || the next tap's W_hi -> W_A, X_hi -> the set X_lo(kw) has left
            if kw < 2:
                nxt = wreads(0, kw + 1, "A") + xreads(0, kw + 1, l)
                ops = sched("B", h, False, nxt, per[3 * kw + 2], None, xands(kw + 1, l, 0))
                pend = linearize(e, ops, pend)
        h2, l2 = hi(2), 1 - hi(2)
        if mode == "last":
            linearize(e, sched("B", h2, False, [], [], None, []), pend)
            return
        e(f"s_waitcnt vmcnt({(NP if mode == 'full' else 0) + r_prev})")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        advance(e)
        req = [("x", ln) for ln in res_request(res_tile, kind)] if res_tile is not None else []
        nxt = wreads(0, 0, "A") + xreads(0, 0, l2)              # the NEXT stage's group (0, 0): its parity is 1 - p = l2
        pend = linearize(e, sched("B", h2, False, nxt, req, None, xands(0, l2, 0)), [])
        assert l2 == 1 - p and pend == LP[1 - p], (pend, LP[1 - p])

    LOOP, REST = e.lab("loop"), e.lab("rest")
    for k in range(UNROLL_PAIR):
        body(k & 1, "full", first=(k == 0), res_tile=(k if k < NTILES else None), r_prev=(R if 0 < k <= NTILES else 0))
    e(f"s_sub_u32 s85, s85, {UNROLL_PAIR}")
    e("s_cmp_eq_u32 s85, 0")
    e(f"s_cbranch_scc1 {REST}")
    e.label(LOOP)
    body(0, "full")
    body(1, "full")
    e("s_sub_u32 s85, s85, 2")
    e("s_cmp_lg_u32 s85, 0")
    e(f"s_cbranch_scc1 {LOOP}")
    e.label(REST)
    body(0, "nodma")
    body(1, "last")
    for _ in range(3):
        e("s_nop 7")


def spread_keep(ops, extras):
    """Insert op lists `extras` after MFMAs 2, 5, 8, 11 ... of an already interleaved op list."""
    if not extras:
        return ops
    idx = [k for k, op in enumerate(ops) if op[0] == "m" and "v_mfma" in op[1]]
    n = len(idx)
    slots = {}
    for k, ex in enumerate(extras):
        pos = idx[min(n - 1, 1 + (k * n) // len(extras))]
        slots.setdefault(pos, []).extend(ex)
    out = []
    for k, op in enumerate(ops):
        out.append(op)
        out.extend(slots.get(k, []))
    return out


# ---------------------------------------------------------------- epilogue
NTILES = NI * NJ


BIAS_LDS = 384                                     # the wave's 96 bias values sit behind its 96 gamma values in LDS


def bias_runs(e):
    """The lane's bias runs (8 couts 32 i + 16 p + 8 h ..) -> v[BIASV ...]: from the wave's block in LDS
    ([gamma(96) | bias(96)] fp32 above the stages, written by the C++ prologue: zeros where there is no bias / no such
    cout).  Round 5: they were twelve buffer loads issued here and waited for at once — ~1.5 us of exposed memory latency
    per tile with the matrix pipe idle."""
    e(f"v_add_u32 v{V_ST_OFF}, {S_GAMMA_LDS}, v{V_LANE}")         # (v249 is free until the row-block loop sets it)
    for i in range(NI):
        for p in range(2):
            for q in range(2):
                e(f"ds_read_b128 {vr(BIASV + (2 * i + p) * 8 + 4 * q, 4)}, v{V_ST_OFF} "
                  f"offset:{BIAS_LDS + (32 * i + 16 * p) * 4 + 16 * q}")


def epilogue(e, kind):
    """y = acc + bias (+ residual), bf16 or fp32 (wide_epilogue's order).  After the permlane widening lane (r, h)
    holds, per tile (i, j) and run p, the 8 couts 32 i + 16 p + 8 h .. of voxel row 32 j + r of the wave's patch.
    The residual tiles were requested during the k loop (res_request): nothing to wait for but the bias."""
    es = 2 if kind == "bf16" else 4
    e(f"v_mbcnt_lo_u32_b32 v{V_LANE}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 v{V_LANE}, -1, v{V_LANE}")             # lane
    e(f"v_lshrrev_b32 v{V_LANE}, 5, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_LANE}, 5, v{V_LANE}")                   # 32 h bytes = 8 h floats
    bias_runs(e)
    e(f"v_mov_b32 v{V_ST_OFF}, v{VOC}")                           # store offset of the lane's row in row block j
    for n in range(NTILES):
        j, i = divmod(n, NI)
        t = i * NJ + j
        if i == 0 and j:
            e(f"v_add_u32 v{V_ST_OFF}, {S_JSTEP}, v{V_ST_OFF}")
        for r_ in range(16):
            e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
        e("s_nop 1")
        for q0 in (0, 2):                                        # quads (0,1), (2,3) -> two runs of 8 consecutive couts
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        bank, rbase = res_slot(n, kind)
        if bank == "a":                                          # parked in AGPRs: into the slot tile n - 8 has left
            for r_ in range(16):
                e(f"v_accvgpr_read_b32 v{120 + 16 * (n - 8) + r_}, a{rbase + r_}")
            rbase = 120 + 16 * (n - 8)
        if n == 0:
            e("s_waitcnt vmcnt(0)")                              # every residual tile (requested during the k loop)
            e("s_waitcnt lgkmcnt(0)")                            # the bias runs (LDS)
        e(f"v_bfe_u32 v{V_ROWBIT}, v{ROWMASK}, {j}, 1")           # this lane's row of the strip is an output row
        e(f"v_cmp_ne_u32 vcc, 0, v{V_ROWBIT}")
        e("s_and_saveexec_b64 s[86:87], vcc")
        for p in range(2):
            v0 = T + 8 * p
            b0 = BIASV + (2 * i + p) * 8
            r0 = rbase + (p * 4 if kind == "bf16" else p * 8)
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(b0 + r_, 2)}")
            if kind == "bf16":
                for r_ in range(4):                              # residual bf16 pairs -> fp32, added after the bias
                    e(f"v_lshlrev_b32 v{V_PAIR}, 16, v{r0 + r_}")
                    e(f"v_and_b32 v{V_PAIR + 1}, 0xffff0000, v{r0 + r_}")
                    e(f"v_pk_add_f32 {vr(v0 + 2 * r_, 2)}, {vr(v0 + 2 * r_, 2)}, v[{V_PAIR}:{V_PAIR + 1}]")
                for r_ in range(4):
                    e(f"v_cvt_pk_bf16_f32 v{v0 + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
                e(f"buffer_store_dwordx4 {vr(v0, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es}")
            else:
                for r_ in range(0, 8, 2):
                    e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(r0 + r_, 2)}")
                e(f"buffer_store_dwordx4 {vr(v0, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es}")
                e(f"buffer_store_dwordx4 {vr(v0 + 4, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es + 16}")
        e("s_nop 1")
        e("s_mov_b64 exec, s[86:87]")


V_SS, V_INV, V_GADDR, V_NOFF = 254, 251, 248, 250
RNORM = 216                                       # bf16 norm kind: the row's three packed tiles, v[216:239]; v[240:247] gamma
S_GAMMA_LDS, S_SQRTC = S_SPARE1, S_SPARE2


def silu_norm(e, regs, gam, inv):
    """regs[k] <- silu((inv * regs[k]) * gam[k]) in rms_silu_kernel's operation order (vae_elementwise.hip); values in
    pairs so that a transcendental's result is never read by the next instruction (gfx950 trans forwarding: 1 wait)."""
    for k in range(0, len(regs), 2):
        a, g = regs[k:k + 2], gam[k:k + 2]
        t = (V_PAIR, V_PAIR + 1)
        for x in range(2):
            e(f"v_mul_f32 v{a[x]}, v{inv}, v{a[x]}")
        for x in range(2):
            e(f"v_mul_f32 v{a[x]}, v{a[x]}, v{g[x]}")
        for x in range(2):
            e(f"v_mul_f32 v{t[x]}, 0xbfb8aa3b, v{a[x]}")         # -log2(e) x
        for x in range(2):
            e(f"v_exp_f32 v{t[x]}, v{t[x]}")
        for x in range(2):
            e(f"v_add_f32 v{t[x]}, 1.0, v{t[x]}")
        for x in range(2):
            e(f"v_rcp_f32 v{t[x]}, v{t[x]}")
        for x in range(2):
            e(f"v_mul_f32 v{a[x]}, v{a[x]}, v{t[x]}")


def epilogue_norm(e, kind, pair_out=False):
    """epilogue() followed, per row block j, by the NEXT layer's RMS norm + SiLU of the 96 output channels (vae.py:39-54
    under :195-197 / :203-205): bf16 [voxel][96] to %[rnorm].  A voxel row's 96 values sit in two lanes (r, h = 0 / 1),
    48 each: sums of squares as one fma chain per run p (channel order, like rms_silu_kernel's lane p + 2... see there),
    the two runs added, the halves exchanged with v_permlane32_swap, then sqrt / max / rcp / mul and the SiLU with the
    same instructions in the same order as the stand-alone kernel: bit-identical outputs.  gamma is read from LDS
    (96 floats above the stages, written by the C++ prologue).  fp32 kind: y stays in the residual slot of its tile
    until the row is complete; bf16 kind: the rounded y (what the stand-alone kernel would read back) as packed pairs."""
    es = 2 if kind == "bf16" else 4
    e(f"v_mbcnt_lo_u32_b32 v{V_LANE}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 v{V_LANE}, -1, v{V_LANE}")
    e(f"v_lshrrev_b32 v{V_LANE}, 5, v{V_LANE}")
    e(f"v_lshlrev_b32 v{V_LANE}, 5, v{V_LANE}")                   # 32 h bytes = 8 h floats
    bias_runs(e)
    e(f"v_add_u32 v{V_GADDR}, {S_GAMMA_LDS}, v{V_LANE}")          # this lane's gamma runs in LDS
    e(f"v_mov_b32 v{V_ST_OFF}, v{VOC}")
    for n in range(NTILES):
        j, i = divmod(n, NI)
        t = i * NJ + j
        if i == 0 and j:
            e(f"v_add_u32 v{V_ST_OFF}, {S_JSTEP}, v{V_ST_OFF}")
        for r_ in range(16):
            e(f"v_accvgpr_read_b32 v{T + r_}, a{t * 16 + r_}")
        e("s_nop 1")
        for q0 in (0, 2):
            for r_ in range(4):
                e(f"v_permlane32_swap_b32 v{T + 4 * q0 + r_}, v{T + 4 * (q0 + 1) + r_}")
        bank, rbase = res_slot(n, kind)
        if bank == "a":
            for r_ in range(16):
                e(f"v_accvgpr_read_b32 v{120 + 16 * (n - 8) + r_}, a{rbase + r_}")
            rbase = 120 + 16 * (n - 8)
        if n == 0:
            e("s_waitcnt vmcnt(0)")
            e("s_waitcnt lgkmcnt(0)")                            # the bias runs (LDS)
        e(f"v_bfe_u32 v{V_ROWBIT}, v{ROWMASK}, {j}, 1")
        e(f"v_cmp_ne_u32 vcc, 0, v{V_ROWBIT}")
        e("s_and_saveexec_b64 s[86:87], vcc")
        for p in range(2):
            v0 = T + 8 * p
            b0 = BIASV + (2 * i + p) * 8
            ss = V_SS + p
            first = "v_mul_f32 v{0}, v{1}, v{1}" if i == 0 else "v_fmac_f32 v{0}, v{1}, v{1}"
            for r_ in range(0, 8, 2):
                e(f"v_pk_add_f32 {vr(v0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(b0 + r_, 2)}")
            if kind == "bf16":
                r0 = rbase + p * 4
                for r_ in range(4):
                    e(f"v_lshlrev_b32 v{V_PAIR}, 16, v{r0 + r_}")
                    e(f"v_and_b32 v{V_PAIR + 1}, 0xffff0000, v{r0 + r_}")
                    e(f"v_pk_add_f32 {vr(v0 + 2 * r_, 2)}, {vr(v0 + 2 * r_, 2)}, v[{V_PAIR}:{V_PAIR + 1}]")
                keep = RNORM + 8 * i + 4 * p                     # the rounded y, packed: kept for the row's norm
                for r_ in range(4):
                    e(f"v_cvt_pk_bf16_f32 v{keep + r_}, v{v0 + 2 * r_}, v{v0 + 2 * r_ + 1}")
                e(f"buffer_store_dwordx4 {vr(keep, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es}")
                for r_ in range(4):
                    e(f"v_lshlrev_b32 v{V_PAIR}, 16, v{keep + r_}")
                    e(f"v_and_b32 v{V_PAIR + 1}, 0xffff0000, v{keep + r_}")
                    e((first if r_ == 0 else "v_fmac_f32 v{0}, v{1}, v{1}").format(ss, V_PAIR))
                    e(f"v_fmac_f32 v{ss}, v{V_PAIR + 1}, v{V_PAIR + 1}")
            else:
                r0 = rbase + p * 8                               # y = (acc + bias) + residual, left in the residual's slot
                for r_ in range(0, 8, 2):
                    e(f"v_pk_add_f32 {vr(r0 + r_, 2)}, {vr(v0 + r_, 2)}, {vr(r0 + r_, 2)}")
                e(f"buffer_store_dwordx4 {vr(r0, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es}")
                e(f"buffer_store_dwordx4 {vr(r0 + 4, 4)}, v{V_ST_OFF}, %[ry], 0 offen offset:{(i * 32 + 16 * p) * es + 16}")
                for r_ in range(8):
                    e((first if r_ == 0 else "v_fmac_f32 v{0}, v{1}, v{1}").format(ss, r0 + r_))
        e("s_nop 1")
        e("s_mov_b64 exec, s[86:87]")
        if i != NI - 1:
            continue
        # ---- the row block is complete: 1 / rms of every voxel row, then norm + SiLU of its three tiles
        e(f"v_add_f32 v{V_SS}, v{V_SS}, v{V_SS + 1}")
        e(f"v_mov_b32 v{V_SS + 1}, v{V_SS}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 v{V_SS}, v{V_SS + 1}")          # v254 = the low half's sum, v255 = the high half's, in all lanes
        e("s_nop 1")
        e(f"v_add_f32 v{V_SS}, v{V_SS}, v{V_SS + 1}")
        e(f"v_sqrt_f32 v{V_SS}, v{V_SS}")
        e("s_nop 0")
        e(f"v_max_f32 v{V_SS}, 0x2b8cbccc, v{V_SS}")              # 1e-12
        e(f"v_rcp_f32 v{V_SS}, v{V_SS}")
        e("s_nop 0")
        e(f"v_mul_f32 v{V_INV}, {S_SQRTC}, v{V_SS}")
        e(f"v_bfe_u32 v{V_ROWBIT}, v{ROWMASK}, {j}, 1")
        e(f"v_cmp_ne_u32 vcc, 0, v{V_ROWBIT}")
        e("s_and_saveexec_b64 s[86:87], vcc")
        noff = V_ST_OFF
        if pair_out:
            # split-bf16 PAIR output [voxel][2 x 96] (omh_conv_args.pair: the next convolution's input): 384 bytes per
            # voxel like the fp32 y, the lane's 8 channels of block 2 i + p at + 16 h instead of + 32 h
            # (32 h from the gamma address: v251 = V_LANE has become V_INV by now)
            e(f"v_subrev_u32 v{V_NOFF}, {S_GAMMA_LDS}, v{V_GADDR}")
            e(f"v_lshrrev_b32 v{V_NOFF}, 1, v{V_NOFF}")
            e(f"v_sub_u32 v{V_NOFF}, v{V_ST_OFF}, v{V_NOFF}")
            noff = V_NOFF
        elif kind != "bf16":
            e(f"v_lshrrev_b32 v{V_NOFF}, 1, v{V_ST_OFF}")         # bf16 output: half the fp32 output's byte offset
            noff = V_NOFF
        for i2 in range(NI):
            n2 = NI * j + i2
            if kind == "bf16":
                for p in range(2):
                    keep = RNORM + 8 * i2 + 4 * p
                    for q in range(2):
                        e(f"ds_read_b128 {vr(240 + 4 * q, 4)}, v{V_GADDR} offset:{(32 * i2 + 16 * p) * 4 + 16 * q}")
                    for r_ in range(4):
                        e(f"v_lshlrev_b32 v{T + 2 * r_}, 16, v{keep + r_}")
                        e(f"v_and_b32 v{T + 2 * r_ + 1}, 0xffff0000, v{keep + r_}")
                    e("s_waitcnt lgkmcnt(0)")
                    silu_norm(e, [T + r_ for r_ in range(8)], [240 + r_ for r_ in range(8)], V_INV)
                    for r_ in range(4):
                        e(f"v_cvt_pk_bf16_f32 v{T + r_}, v{T + 2 * r_}, v{T + 2 * r_ + 1}")
                    e(f"buffer_store_dwordx4 {vr(T, 4)}, v{noff}, %[rnorm], 0 offen offset:{(i2 * 32 + 16 * p) * 2}")
                    e("s_nop 1")                                 # the store's data registers are rewritten next
            else:
                _, sb = res_slot(n2, kind)
                sr = sb if n2 < 8 else 120 + 16 * (n2 - 8)
                for p in range(2):
                    for q in range(2):
                        e(f"ds_read_b128 {vr(T + 8 * p + 4 * q, 4)}, v{V_GADDR} offset:{(32 * i2 + 16 * p) * 4 + 16 * q}")
                e("s_waitcnt lgkmcnt(0)")
                silu_norm(e, [sr + r_ for r_ in range(16)], [T + r_ for r_ in range(16)], V_INV)
                if pair_out:
                    # hi = bf16(v) -> v[T:T+7], lo = bf16(v - hi) -> v[T+8:T+15] (omh_rms_silu_cl_pair's operations)
                    for r_ in range(8):
                        e(f"v_cvt_pk_bf16_f32 v{T + r_}, v{sr + 2 * r_}, v{sr + 2 * r_ + 1}")
                    for r_ in range(8):
                        e(f"v_lshlrev_b32 v{V_PAIR}, 16, v{T + r_}")
                        e(f"v_and_b32 v{V_PAIR + 1}, 0xffff0000, v{T + r_}")
                        e(f"v_sub_f32 v{sr + 2 * r_}, v{sr + 2 * r_}, v{V_PAIR}")
                        e(f"v_sub_f32 v{sr + 2 * r_ + 1}, v{sr + 2 * r_ + 1}, v{V_PAIR + 1}")
                        e(f"v_cvt_pk_bf16_f32 v{T + 8 + r_}, v{sr + 2 * r_}, v{sr + 2 * r_ + 1}")
                    for p in range(2):
                        blk = 2 * i2 + p
                        e(f"buffer_store_dwordx4 {vr(T + 4 * p, 4)}, v{noff}, %[rnorm], 0 offen offset:{blk * 64}")
                        e(f"buffer_store_dwordx4 {vr(T + 8 + 4 * p, 4)}, v{noff}, %[rnorm], 0 offen offset:{blk * 64 + 32}")
                    e("s_nop 1")                                 # the stores' data registers are rewritten by the next tile
                    continue
                for r_ in range(8):
                    e(f"v_cvt_pk_bf16_f32 v{sr + r_}, v{sr + 2 * r_}, v{sr + 2 * r_ + 1}")
                for p in range(2):
                    e(f"buffer_store_dwordx4 {vr(sr + 4 * p, 4)}, v{noff}, %[rnorm], 0 offen offset:{(i2 * 32 + 16 * p) * 2}")
        e("s_nop 1")
        e("s_mov_b64 exec, s[86:87]")


def generate(cfg, kind, norm=False, pair=False):
    e = Emit(cfg + kind + ("n" if norm else "") + ("p" if pair else ""))
    NA, NB = CONFIGS[cfg]
    (main_loop_pair if pair else main_loop)(e, NA, NB, kind)
    if norm:
        epilogue_norm(e, kind, pair_out=pair)
    else:
        epilogue(e, kind)
    return e


def main():
    print("// GENERATED by gen_conv_w64.py — do not edit; edit the generator.")
    for cfg in CONFIGS:
        for kind in KINDS:
            for norm in ((False, True) if cfg == "P" else (False,)):    # the fused norm needs all 96 channels in one wave
                e = generate(cfg, kind, norm)
                print(f"#define OMH_CONV_W64_ASM_{cfg}_{kind.upper()}{'_NORM' if norm else ''} \\")
                print(" \\\n".join(e.text().split("\n")))
                print("")
                print(f"// {cfg} {kind}{' norm' if norm else ''}: {len(e.lines)} lines, {sum('v_mfma' in ln for ln in e.lines)} MFMA")
    for cfg, norm in (("P", False), ("P", True), ("Q", False)):   # the split-bf16 pair streams (fp32-faithful VAE mode): fp32 out
        e = generate(cfg, "f32", norm, pair=True)
        print(f"#define OMH_CONV_W64_ASM_{cfg}_F32_PAIR{'_NORM' if norm else ''} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        print(f"// {cfg} f32 pair{' norm' if norm else ''}: {len(e.lines)} lines, {sum('v_mfma' in ln for ln in e.lines)} MFMA")
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(80, 100)] + [f'"v{i}"' for i in range(12, 256)] + \
           [f'"a{i}"' for i in range(256)]
    print("#define OMH_CONV_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))


if __name__ == "__main__":
    main()
