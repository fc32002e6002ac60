#!/usr/bin/env python3
"""Generator of the instruction streams of ``gemm_bf16_tn_w64_kernel`` (gemm_tn_w64.hip): C[M, N] (+)= A[K, M]^T B[K, N],
both operands row-major with the contraction index on the ROWS (the weight gradient dW = dy^T x of the training step,
seaweed_apt/distilled_trainer.py:239 `loss.backward()` through every nn.Linear of wan/modules/model.py), on a
256(m) x 384(n) x 64(k) workgroup tile — 4 waves, ONE per SIMD, each a 128 x 192 patch with 384 fp32 accumulators (256
in AGPRs + 128 in arch VGPRs): the k-major sibling of gen_gemm_w64.py (same register map, same two 80 KiB stages, same
step structure), replacing gemm_tn.hip's 128 x 128 tiles (64 flop per staged byte, 0.23 of the MFMA peak measured on a
block's group) with 153.6 flop per staged byte.

    python gen_gemm_tn_w64.py > gemm_tn_w64_asm.inc

LDS tiles (as gemm_tn.hip): A tile [64 k][256 m] bf16 (row pitch 512 B = 32 slots of 16 B), B tile [64 k][384 n] (768 B =
48 slots); physical slot s of row k holds logical slot s ^ ((k & 3) << 2) (applied on the LDS-DMA source address), so
the four k rows of a ds_read_b64_tr_b16 lane group fall on different banks.  A fragment of v_mfma_f32_32x32x16_bf16 is
two transposing reads (k0 .. k0+3 and k0+4 .. k0+7 of the lane's k group).  The m side goes in the MFMA A slot and the n
side in the B slot: a lane ends up with column n = lane & 31 of 16 rows m, so a store instruction writes two 128-byte
runs.

LDS-DMA pieces (1 KiB = 64 lanes x 16 B): wave w stages k rows 16 w .. 16 w + 15 of both tiles — A: 8 pieces of two rows
(two per-lane offset patterns: (row & 3) = 0,1 / 2,3); B: 12 pieces, 1 KiB = 1 1/3 rows, three patterns repeating every
4 rows.  Rows >= K read 0 through the descriptors; columns >= N are marked out of range per lane by the caller (vob*).

Epilogue: "st" C = acc, "acc" C = C + acc (old values requested one accumulator tile ahead).  Columns >= N masked with
EXEC; M % 256 == 0 is the caller's precondition.

Register map: a[0:255] accumulator tiles 0..15, v[128:255] tiles 16..23 (tile = 4 i + j, i = n tile, j = m tile);
k loop: v[32:71] / v[72:111] fragment buffers, v[12:17] / v[22:27] B-fragment addresses of stage 0 / 1, v[18:21] /
v[28:31] A-fragment addresses; epilogue: v[32:47] tile values, v[48:79] two slots of old C values, v[112:116] lane constants / temporaries; s[60:73] scalar
arguments, s[80:95] scratch.
"""
NI, NJ = 6, 4                   # n tiles (B) x m tiles (A) per wave
STAGE = 81920
BOFF = 32768                    # B tile behind the A tile inside a stage
PA, PB = 512, 768               # row pitches
FB = 4 * NI + 4 * NJ
NTILES = NI * NJ
import os
READ_END = int(os.environ.get("OMH_GTW64_READ_END", "20"))       # the next group's 20 reads go behind the first READ_END MFMAs
DMA_START = int(os.environ.get("OMH_GTW64_DMA_START", "4"))      # a group's 6 B pieces: every other MFMA from this one on
# (measured on a block's group, one launch: start 12 -> 203 us, 8 -> 201, 4 -> 197, 2 -> 196, 0 -> 196.5; reads behind the first
# 16 / 20 / 24 MFMAs: 208 / 203 / 204)

S_LDA, S_LDB, S_SAB, S_SBB, S_SAS, S_SBS, S_NK, S_SCB = "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67"
S_LDC4, S_NREM, S_KSA, S_KSB, S_CM, S_CN = "s68", "s69", "s70", "s71", "s72", "s73"
S_KA, S_KB = "s80", "s92"       # running k byte offsets of the A / B tile being fetched


def acc(i, j):
    t = i * NJ + j
    return (f"a[{t * 16}:{t * 16 + 15}]") if t < 16 else (f"v[{128 + (t - 16) * 16}:{128 + (t - 16) * 16 + 15}]")


def nfrag(buf, i):   return 32 + buf * FB + 4 * i
def mfrag(buf, j):   return 32 + buf * FB + 4 * NI + 4 * j
def NA(s, i):        return 12 + s * 10 + i
def MA(s, j):        return 12 + s * 10 + NI + j
def vr(lo, n=1):     return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


class Emit:
    def __init__(self, tag):
        self.lines, self.tag = [], tag

    def __call__(self, s):
        self.lines.append(s)

    def lab(self, name):
        return f".Lgtw64{self.tag}_{name}_%="

    def label(self, name):
        self.lines.append(f"{name}:")

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


def linearize(e, ops, pending):
    """Emit `ops` in order; before an MFMA whose fragments are still in flight, the exact lgkmcnt (LDS reads return in
    order; the counter has 4 bits, so more than 15 younger reads wait for a few of them too)."""
    pending = list(pending)
    for op in ops:
        if op[0] == "r":
            e(op[2])
            pending.append(op[1])
        elif op[0] == "m":
            need = [t for t in op[2] if t in pending]
            if need:
                last = max(len(pending) - 1 - pending[::-1].index(t) for t in need)
                allowed = min(15, len(pending) - last - 1)
                e(f"s_waitcnt lgkmcnt({allowed})")
                pending = pending[len(pending) - allowed:]
            e(op[1])
        else:
            e(op[1])
    return pending


def frag_reads(stage, kk, buf):
    """The 2 (NI + NJ) transposing reads of k group kk, in the order the MFMAs want them: B0, A0..A3, B1..B5."""
    def one(name, reg, addr, pitch):
        return [("r", name, f"ds_read_b64_tr_b16 {vr(reg, 2)}, {vr(addr)} offset:{kk * 16 * pitch}"),
                ("r", name, f"ds_read_b64_tr_b16 {vr(reg + 2, 2)}, {vr(addr)} offset:{kk * 16 * pitch + 4 * pitch}")]
    out = one(f"N{buf}.0", nfrag(buf, 0), NA(stage, 0), PB)
    for j in range(NJ):
        out += one(f"M{buf}.{j}", mfrag(buf, j), MA(stage, j), PA)
    for i in range(1, NI):
        out += one(f"N{buf}.{i}", nfrag(buf, i), NA(stage, i), PB)
    return out


def group_mfmas(buf, first=False):
    out = []
    for i in range(NI):
        for j in range(NJ):
            c = "0" if first else acc(i, j)
            out.append(("m", f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {vr(mfrag(buf, j), 4)}, {vr(nfrag(buf, i), 4)}, {c}",
                        [f"N{buf}.{i}", f"M{buf}.{j}"]))
    return out


def dma_piece(stage, operand, q):
    """One 1 KiB LDS-DMA piece of this wave's 16 k rows.  A: rows 2 q, 2 q + 1; B: slots 64 q .. 64 q + 63 of the 16 x 48.
    s81 / s82: running source offsets.  The rows of a partial last k tile are out of range through the descriptors:
    gfx950 range-checks voffset + inst_offset + soffset without 32-bit wrap-around (tools/probes/buffer_range_probe.hip:
    a soffset alone past num_records reads 0 and drops stores)."""
    if operand == "a":
        out = [f"s_add_u32 m0, {S_LDA}, {stage * STAGE + q * 1024}"]
        out.append(f"s_add_u32 s81, {S_SAB}, {S_KA}" if q == 0 else f"s_add_u32 s81, s81, {S_SAS}")
        out.append("buffer_load_dwordx4 %[voa" + str(q & 1) + "], %[ra], s81 offen lds")
    else:
        out = [f"s_add_u32 m0, {S_LDB}, {stage * STAGE + q * 1024}"]
        if q == 0:
            out.append(f"s_add_u32 s82, {S_SBB}, {S_KB}")
        elif q % 3 == 0:
            out.append(f"s_add_u32 s82, s82, {S_SBS}")
        else:
            out.append("s_nop 0")                                # one wait state between the M0 write and the LDS-DMA
        out.append("buffer_load_dwordx4 %[vob" + str(q % 3) + "], %[rb], s82 offen lds")
    return [("x", ln) for ln in out]


def all_pieces(stage):
    return [dma_piece(stage, "a", q) for q in range(8)] + [dma_piece(stage, "b", q) for q in range(12)]


def tile_prologue(e):
    """k tile 0 -> stage 0 (all 20 pieces per wave), k tile 1 -> stage 1 (the 8 A pieces; the k loop issues the B ones)."""
    e(f"s_mov_b32 {S_KA}, 0")
    e(f"s_mov_b32 {S_KB}, 0")
    for ops in all_pieces(0):
        for op in ops:
            e(op[1])
    e(f"s_mov_b32 {S_KA}, {S_KSA}")
    e(f"s_mov_b32 {S_KB}, {S_KSB}")
    for ops in all_pieces(1)[:8]:
        for op in ops:
            e(op[1])


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


def main_loop(e):
    """One output tile: prologue DMA, the k loop (>= 3 k tiles), then every wave is done with the LDS stages."""
    tile_prologue(e)
    e("v_mbcnt_lo_u32_b32 v114, -1, 0")
    e("v_mbcnt_hi_u32_b32 v114, -1, v114")                      # lane
    e("v_bfe_u32 v115, v114, 2, 2")                             # fe = (lane >> 2) & 3: k row of the lane inside its group of 4
    e("v_and_b32 v116, 31, v114")                               # column of the lane inside a 32 x 32 tile
    for j in range(NJ):
        e(f"s_add_u32 s90, {S_CM}, {j}")
        e("v_xor_b32 v112, s90, v115")
        e(f"v_lshl_add_u32 {vr(MA(0, j))}, v112, 6, %[mab]")
        e(f"v_add_u32 {vr(MA(1, j))}, {STAGE}, {vr(MA(0, j))}")
    for i in range(NI):
        e(f"s_add_u32 s90, {S_CN}, {i}")
        e("v_xor_b32 v112, s90, v115")
        e(f"v_lshl_add_u32 {vr(NA(0, i))}, v112, 6, %[nab]")
        e(f"v_add_u32 {vr(NA(1, i))}, {STAGE}, {vr(NA(0, i))}")
    e("s_waitcnt vmcnt(8)")                                     # k tile 0 has landed (in-order counter; the previous
    e("s_barrier")                                              # tile's stores, older still, too)
    pend = linearize(e, frag_reads(0, 0, 0), [])
    LOOP_PENDING = list(pend)
    LOOP, DONE = e.lab("loop"), e.lab("done")
    NM = NI * NJ

    def body(s, first=False, mode="full"):
        """One k step on stage s.  Groups 0..2: MFMAs || reads of the next group || the 12 B pieces of k tile kt+1 (into
        stage s^1).  Then everything in flight is waited for, barrier, and group 3 runs || the 8 A pieces of k tile kt+2
        (into stage s, free now) || reads of group 0 of stage s^1.  mode "nox": the step before the last, "none": the last."""
        pend = LOOP_PENDING
        rest = all_pieces(s ^ 1)[8:] if mode != "none" else []
        for kk in range(3):
            mf = group_mfmas(kk & 1, first=(first and kk == 0))
            reads = frag_reads(s, kk + 1, (kk + 1) & 1)
            reads = [reads[k:k + 2] for k in range(0, len(reads), 2)]
            ops = spread_after(mf, reads, 0, READ_END)
            if kk < 2:
                ops = with_dma_tail(ops, rest[kk * 6:(kk + 1) * 6], DMA_START)
            pend = linearize(e, ops, pend)
        if mode == "none":
            linearize(e, group_mfmas(1), pend)
            return
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        xp = []
        if mode == "full":
            e(f"s_add_u32 {S_KA}, {S_KA}, {S_KSA}")             # k offsets of tile kt+2
            e(f"s_add_u32 {S_KB}, {S_KB}, {S_KSB}")
            xp = all_pieces(s)[:8]
        reads = frag_reads(s ^ 1, 0, 0)
        reads = [reads[k:k + 2] for k in range(0, len(reads), 2)]
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
    body(0, first=True)                                         # k step 0 (>= 3 k tiles: never a tail step)
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


T, CO = 32, 48                  # v[32:47] tile values; v[48:63], v[64:79] old C values of two tiles in flight


def row_soffset(e, j, r):
    """s88 = byte offset of row 32 j + 8 (r >> 2) + (r & 3) of the workgroup's tile (the lane's 4 h is in voc)."""
    e(f"s_mul_i32 s88, {S_LDC4}, {32 * j + 8 * (r >> 2) + (r & 3)}")
    e(f"s_add_u32 s88, s88, {S_SCB}")


def epilogue(e, accumulate):
    """Tile (i, j), register r of a lane = C[32 j + 8 (r >> 2) + 4 h + (r & 3)][32 i + (lane & 31)] of the wave's patch."""
    order = [(i, j) for i in range(NI) for j in range(NJ)]

    def mask(i):
        e(f"v_add_u32 v113, {32 * i}, v116")
        e(f"v_cmp_gt_u32 vcc, {S_NREM}, v113")
        e("s_and_saveexec_b64 s[86:87], vcc")

    def unmask():
        e("s_nop 1")
        e("s_mov_b64 exec, s[86:87]")

    def loads(k):
        i, j = order[k]
        mask(i)
        for r in range(16):
            row_soffset(e, j, r)
            e(f"buffer_load_dword v{CO + 16 * (k & 1) + r}, %[voc], %[rc], s88 offen offset:{128 * i}")
        unmask()

    if accumulate:
        loads(0)
    for k, (i, j) in enumerate(order):
        t = i * NJ + j
        if accumulate and k + 1 < len(order):
            loads(k + 1)
        if t < 16:
            for r in range(16):
                e(f"v_accvgpr_read_b32 v{T + r}, a{t * 16 + r}")
            src = T
        else:
            src = 128 + (t - 16) * 16
        if accumulate:
            # in order: loads k | stores k-1 | loads k+1 -> tile k's values are there once 32 (16 at the end) are left
            e(f"s_waitcnt vmcnt({(16 if k + 1 < len(order) else 0) + (16 if k > 0 else 0)})")
            for r in range(16):
                e(f"v_add_f32 v{T + r}, v{CO + 16 * (k & 1) + r}, v{src + r}")
            src = T
        else:
            e("s_nop 1")
        mask(i)
        for r in range(16):
            row_soffset(e, j, r)
            e(f"buffer_store_dword v{src + r}, %[voc], %[rc], s88 offen offset:{128 * i}")
        unmask()


def generate(accumulate):
    e = Emit("acc" if accumulate else "st")
    main_loop(e)
    epilogue(e, accumulate)
    return e


def main():
    print("// GENERATED by gen_gemm_tn_w64.py — do not edit; edit the generator.")
    for name, e in (("ST", generate(False)), ("ACC", generate(True))):
        print(f"#define OMH_GEMM_TN_W64_ASM_{name} \\")
        print(" \\\n".join(e.text().split("\n")))
        print("")
        print(f"// {name}: {len(e.lines)} lines, {sum('v_mfma' in ln for ln in e.lines)} MFMA")
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(80, 96)] + [f'"v{i}"' for i in range(12, 256)] + \
           [f'"a{i}"' for i in range(256)]
    print("#define OMH_GEMM_TN_W64_CLOBBERS \\")
    rows = [", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12)]
    print("    " + ", \\\n    ".join(rows))


if __name__ == "__main__":
    main()
