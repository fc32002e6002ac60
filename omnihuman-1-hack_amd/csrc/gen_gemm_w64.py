#!/usr/bin/env python3
"""Generator of the instruction stream of ``gemm_bf16_nt_w64_kernel`` (gemm_w64.hip): C = A B^T for the large DiT
GEMMs on a 256(m) x 384(n) x 64 workgroup tile — 4 waves, ONE per SIMD, each owning a 128 x 192 patch with 384
fp32 accumulators (256 in AGPRs + 128 in arch VGPRs).

Why: the 8-wave 256 x 256 kernel (gemm_bf16.hip) is bound by the L2 -> LDS fill stream (22 B/clk/CU);
what is left is bytes per flop, and a 256 x 384 tile moves 17 % fewer (153.6 flop per staged byte against 128).  It
does not fit two waves per SIMD (192 accumulators + fragments per wave > 256), and a compiler-scheduled one-wave
variant lost to its own LDS-DMA issue stalls in round 1 — hence a generated stream, as for the attention kernel
(gen_attn_w64.py): fixed register map, exact lgkmcnt waits, the 20 DMA pieces of a k-step spread between the MFMAs.

    python gen_gemm_w64.py > gemm_w64_asm.inc

Layout (as gemm_bf16.hip): operand tiles [rows][64] bf16 in LDS, 16-byte-slot XOR swizzle ((row >> 1) & 7) applied on
the LDS-DMA source address and on the ds_read_b128; weights (B, n) in the MFMA A slot, activations (A, m) in the B
slot, so a lane ends up with 4 consecutive n of one row m.  Two stages of 80 KiB (X tile 32 KiB | W tile 48 KiB).

Register map: a[0:255] accumulator tiles 0..15, v[128:255] tiles 16..23 (tile = 4 i + j, i = n tile, j = m tile);
v[32:71] / v[72:111] fragment buffers (6 W fragments + 4 X fragments of one 16-wide k group each);
v[12:19] X fragment addresses [stage][kk], v[20:27] W fragment addresses; s80.. scratch.
"""
import sys

NI, NJ = 6, 4                   # n tiles (weights) x m tiles (activations) per wave
STAGE = 81920
WOFF = 32768                    # W tile behind the X tile inside a stage


def acc(i, j):
    t = i * NJ + j
    return (f"a[{t * 16}:{t * 16 + 15}]") if t < 16 else (f"v[{128 + (t - 16) * 16}:{128 + (t - 16) * 16 + 15}]")


def acc_reg(i, j, r):
    t = i * NJ + j
    return f"a{t * 16 + r}" if t < 16 else f"v{128 + (t - 16) * 16 + r}"


def wfrag(buf, i):   return 32 + buf * 40 + 4 * i
def xfrag(buf, j):   return 32 + buf * 40 + 24 + 4 * j
def XA(s, kk):       return 12 + s * 4 + kk
def WA(s, kk):       return 20 + s * 4 + kk
def vr(lo, n=1):     return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def lab(self, name):
        return f".Lgw64_{name}_%="

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
    """10 ds_read_b128 of one 16-wide k group: 6 W fragments, 4 X fragments."""
    out = []
    for i in range(NI):
        out.append(("r", f"W{buf}.{i}", f"ds_read_b128 {vr(wfrag(buf, i), 4)}, {vr(WA(stage, kk))} offset:{i * 4096}"))
    for j in range(NJ):
        out.append(("r", f"X{buf}.{j}", f"ds_read_b128 {vr(xfrag(buf, j), 4)}, {vr(XA(stage, kk))} offset:{j * 4096}"))
    return out


def group_mfmas(buf, first=False):
    """24 MFMAs of one k group, ordered so that consecutive MFMAs never share an accumulator and the W fragments
    (read first) are needed first."""
    out = []
    for i in range(NI):
        for j in range(NJ):
            c = "0" if first else acc(i, j)
            out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(wfrag(buf, i), 4)}, {vr(xfrag(buf, j), 4)}, {c}",
                        [f"W{buf}.{i}", f"X{buf}.{j}"]))
    return out


def dma_piece(stage, operand, q):
    """One 1 KiB LDS-DMA piece.  X: wave w fetches rows w*64 + 8 q (q < 8); W: rows w*96 + 8 q (q < 12).
    s81 / s82: running source offsets of the X / W piece (advanced by the piece stride), s83: LDS piece cursor."""
    base = stage * STAGE + (WOFF if operand == "w" else 0)
    vo = ("%[vow1]" if q & 1 else "%[vow0]") if operand == "w" else ("%[vox1]" if q & 1 else "%[vox0]")
    rs = "%[rb]" if operand == "w" else "%[ra]"
    sreg = "s82" if operand == "w" else "s81"
    lds = "%[ldw]" if operand == "w" else "%[ldx]"
    out = [f"s_add_u32 m0, {lds}, {base + q * 1024}"]
    if q == 0:
        out.append(f"s_add_u32 {sreg}, {'%[swb]' if operand == 'w' else '%[sxb]'}, s80")     # tile base + k offset
    else:
        out.append(f"s_add_u32 {sreg}, {sreg}, {'%[sws]' if operand == 'w' else '%[sxs]'}")
    out.append(f"buffer_load_dwordx4 {vo}, {rs}, {sreg} offen lds")
    return [("x", ln) for ln in out]


def all_pieces(stage):
    out = []
    for q in range(8):
        out.append(dma_piece(stage, "x", q))
    for q in range(12):
        out.append(dma_piece(stage, "w", q))
    return out                                              # 20 lists of 3 lines


def spread_after(mfmas, extras, start=0, end=None):
    """Insert the op lists `extras` (each a list of ops) after MFMAs start..end-1, evenly."""
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


def generate():
    e = Emit()
    # ---------------- prologue: fragment addresses
    for kk in range(4):
        e(f"v_xor_b32 v112, {kk}, %[xh]")
        e(f"v_lshl_add_u32 {vr(XA(0, kk))}, v112, 5, %[xab]")
        e(f"v_lshl_add_u32 {vr(WA(0, kk))}, v112, 5, %[wab]")
        e(f"v_add_u32 {vr(XA(1, kk))}, {STAGE}, {vr(XA(0, kk))}")
        e(f"v_add_u32 {vr(WA(1, kk))}, {STAGE}, {vr(WA(0, kk))}")
    # tile 0 -> stage 0 (all 20 pieces), tile 1 -> stage 1 (first 8 pieces = the X tile; the loop issues the rest)
    e("s_mov_b32 s80, 0")                                       # k byte offset of the tile being fetched
    for ops in all_pieces(0):
        for op in ops:
            e(op[1])
    e("s_mov_b32 s80, 128")
    p1 = all_pieces(1)
    for ops in p1[:8]:
        for op in ops:
            e(op[1])
    e("s_waitcnt vmcnt(8)")                                     # tile 0 has landed
    e("s_barrier")
    pend = linearize(e, frag_reads(0, 0, 0), [])
    e("s_mov_b32 s84, 0")                                       # k step counter
    LOOP_PENDING = list(pend)
    LOOP, DONE = e.lab("loop"), e.lab("done")

    def body(s, first_step_c_zero=False):
        """One k step on stage s.  Groups 0..2: MFMAs || reads of the next group || the 12 W pieces of tile kt+1
        (into stage s^1).  Then everything in flight is waited for, barrier, and group 3 runs || the 8 X pieces of
        tile kt+2 (into stage s, free now) || reads of group 0 of stage s^1."""
        pend = LOOP_PENDING
        rest = all_pieces(s ^ 1)[8:]                            # W pieces of tile kt+1
        for kk in range(3):
            mf = group_mfmas(kk & 1, first=False)
            reads = [[r] for r in frag_reads(s, kk + 1, (kk + 1) & 1)]
            extra = list(reads)
            ops = spread_after(mf, extra, 0, 14)
            if kk < 2:
                ops2 = []
                dm = rest[kk * 6:(kk + 1) * 6]
                # DMA pieces behind the reads: MFMAs 14..23
                idx = 0
                cnt = 0
                for op in ops:
                    ops2.append(op)
                    if op[0] == "m":
                        cnt += 1
                        if cnt > 12 and idx < len(dm) and (cnt - 12) % 2 == 1:
                            ops2.extend(dm[idx]); idx += 1
                while idx < len(dm):
                    ops2.extend(dm[idx]); idx += 1
                ops = ops2
            pend = linearize(e, ops, pend)
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        pend = []
        e("s_barrier")
        e("s_add_u32 s80, s80, 128")                            # next tile's k offset (tile kt+2)
        mf = group_mfmas(1)
        # group 3's fragments (buffer 1) were read during group 2 and are complete (lgkmcnt(0) above)
        xp = all_pieces(s)[:8]
        reads = [[r] for r in frag_reads(s ^ 1, 0, 0)]
        extras = []
        for k in range(max(len(xp), len(reads))):
            if k < len(reads):
                extras.append(reads[k])
            if k < len(xp):
                extras.append(xp[k])
        ops = spread_after(mf, extras, 0, 22)
        pend = linearize(e, ops, pend)
        assert pend == LOOP_PENDING, (pend, LOOP_PENDING)
        e("s_add_u32 s84, s84, 1")
        e("s_cmp_eq_u32 s84, %[nk]")
        e(f"s_cbranch_scc1 {DONE}")

    # first k step: accumulators start from zero (C = 0 on the first MFMA of each tile) — emit a dedicated copy
    def first_body():
        pend = LOOP_PENDING
        rest = all_pieces(1)[8:]
        for kk in range(3):
            mf = group_mfmas(kk & 1, first=(kk == 0))
            reads = [[r] for r in frag_reads(0, kk + 1, (kk + 1) & 1)]
            ops = spread_after(mf, reads, 0, 14)
            if kk < 2:
                dm = rest[kk * 6:(kk + 1) * 6]
                ops2, idx, cnt = [], 0, 0
                for op in ops:
                    ops2.append(op)
                    if op[0] == "m":
                        cnt += 1
                        if cnt > 12 and idx < len(dm) and (cnt - 12) % 2 == 1:
                            ops2.extend(dm[idx]); idx += 1
                while idx < len(dm):
                    ops2.extend(dm[idx]); idx += 1
                ops = ops2
            pend = linearize(e, ops, pend)
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s80, s80, 128")
        xp = all_pieces(0)[:8]
        reads = [[r] for r in frag_reads(1, 0, 0)]
        extras = []
        for k in range(8):
            extras.append(reads[k] if k < len(reads) else [])
            extras.append(xp[k])
        extras += [r for r in reads[8:]]
        pend = linearize(e, spread_after(group_mfmas(1), extras, 0, 22), [])
        assert pend == LOOP_PENDING
        e("s_add_u32 s84, s84, 1")
        e("s_cmp_eq_u32 s84, %[nk]")
        e(f"s_cbranch_scc1 {DONE}")

    first_body()
    e.label(LOOP)
    body(1)
    body(0)
    e(f"s_branch {LOOP}")
    e.label(DONE)
    e("s_waitcnt vmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    # ---------------- epilogue (fp32 C, no bias): lane holds 4 consecutive n of row m per register quad
    # voffset %[voc] = row (m tile 0) * ldc * 4 + 16 h; soffset = column base + j * 32 rows; imm = (32 i + 8 q) * 4
    for j in range(NJ):
        if j == 0:
            e("s_mov_b32 s85, %[scb]")
        else:
            e("s_add_u32 s85, s85, %[scj]")
        for i in range(NI):
            for q in range(4):
                t = i * NJ + j
                if t < 16:
                    src = f"a[{t * 16 + 4 * q}:{t * 16 + 4 * q + 3}]"
                else:
                    b = 128 + (t - 16) * 16 + 4 * q
                    src = f"v[{b}:{b + 3}]"
                e(f"buffer_store_dwordx4 {src}, %[voc], %[rc], s85 offen offset:{i * 128 + q * 32}")
    e("s_waitcnt vmcnt(0)")
    return e


def main():
    e = generate()
    print("// GENERATED by gen_gemm_w64.py — do not edit; edit the generator.")
    print("#define OMH_GEMM_W64_ASM \\")
    print(" \\\n".join(e.text().split("\n")))
    print("")
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(80, 86)] + [f'"v{i}"' for i in range(12, 256)] + \
           [f'"a{i}"' for i in range(256)]
    print("#define OMH_GEMM_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))
    print(f"// {len(e.lines)} lines, {sum('v_mfma' in ln for ln in e.lines)} MFMA")


if __name__ == "__main__":
    main()
