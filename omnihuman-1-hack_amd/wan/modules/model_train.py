"""Training-mode forward/backward of WanModel on the gfx950 kernels — the
student step of seaweed_apt/distilled_trainer.py:241-316 (BASELINE config 3).

The reference gets the backward from autograd over aten ops with one
``torch.utils.checkpoint`` per block (model.py:544-548).  Here autograd only
chains three kinds of hand-written nodes — embed -> 30 x block -> head — so
DDP's bucketed RCCL all-reduce still overlaps with the backward (each block's
parameter gradients become ready when that block's node finishes):

* forward of a block node = the fused inference block (model.py:279-330 on
  libomh.so), saving only its input residual stream (per-block checkpointing,
  as the reference does);
* backward of a block node = recompute the block with un-fused kernels that
  keep the intermediates, then the chain rule by hand: every dgrad / wgrad /
  attention dQ,dK,dV product is ``omh_gemm_bf16`` on transposed bf16 operands,
  surrounded by the kernels of csrc/dit_backward.hip.  Gradients w.r.t. the
  time-embedding vector ``e0``, the text context and ``e`` are accumulated in a
  per-forward state object and consumed by the embed node, which runs last.

Reference quirk kept by default: for ``block_idx > 10``
the reference computes the FFN on the CPU under ``no_grad`` and adds
``0 * ffn_input`` (model.py:317-324), so those FFN weights — and everything
upstream *through that FFN* — receive no gradient; only the gate ``e[5]`` does.
``model.reference_ffn_freeze = False`` turns the quirk off (full gradients).

The attention backward is the fused kernel pair of csrc/attention_bwd.hip (sized for the single-frame training
clips of config 3, S = 1560); ``OMH_ATTN_BWD=unfused`` keeps the first implementation (scores materialised per head
through the GEMM kernel) for A/B runs.  The i2v backbone trains too: the image-token branch of the cross-attention
(k_img / v_img / norm_k_img) and img_emb (LayerNorm, Linear, GELU(erf), Linear, LayerNorm on the CLIP tokens).
"""
import math
import os

import torch

from .._backend import ops

ptr = ops.ptr
EPI_BF16, EPI_F32, EPI_ACC = ops.EPI_BF16, ops.EPI_F32, ops.EPI_F32_ACCUM
BIAS_N, BIAS_M, BIAS_NONE = ops.BIAS_N, ops.BIAS_M, ops.BIAS_NONE


def _ru(a, b):
    return (a + b - 1) // b * b


class _State:
    """Per-forward shared state: the inference context plus gradient accumulators."""

    def __init__(self):
        self.fc = None
        self.d_e0 = self.d_e = self.d_ctx = None


# dgrad products dx = dy W: on a transposed bf16 copy of the weight (one transpose per weight and step, cached per
# weight version) with the row-major-B kernel.  OMH_DGRAD=nn runs them on the weight as stored ([out, in] = k-major
# B of omh_gemm_bf16) instead — no copies, but measured slower (146.1 vs 143.7 ms per 4-clip step: the transposing
# LDS read costs more than the ~250 weight transposes, see gemm_bf16.hip), kept for the tests and for A/B timing.
_DGRAD_NN = os.environ.get("OMH_DGRAD", "nt") == "nn"


def _wT(mod, key, weight_bf16):
    """B operand of the dgrad GEMM for a packed [N, K] weight: its transposed bf16 copy [K, N] (cached per weight
    version), or with OMH_DGRAD=nn the weight itself (k-major B)."""
    if _DGRAD_NN:
        return weight_bf16
    return mod._packed.get("T:" + key, (weight_bf16,), lambda: ops.transpose_bf16(weight_bf16))


def _wT_once(weight_bf16):
    return weight_bf16 if _DGRAD_NN else ops.transpose_bf16(weight_bf16)


# The weight-gradient GEMMs of a block run on a second HIP stream (OMH_WGRAD_STREAM=0: on the main one).  At 4 clips per GPU they are
# tile-poor (1536 x 1536 outputs: 144 tiles with split K) like the dgrad GEMMs they are independent of (150 tiles on 256
# CUs), so the two fill each other's idle CUs.  The side stream forks from the main one at every call (its inputs are
# ready there) and joins at the end of the block's backward (_wgrad_join); hipGraph capture follows the fork / join.
# Measured: 142.7 -> 138.2 ms per step at 4 clips, 428 -> 413 ms at 16, 167 -> 162 ms with every parameter trained.
_WGRAD_STREAM = os.environ.get("OMH_WGRAD_STREAM", "1") == "1"
_side = {}


def _side_stream(dev):
    s = _side.get(dev)
    if s is None:
        s = _side[dev] = torch.cuda.Stream(device=dev)
    return s


def _wgrad_join(dev):
    if _WGRAD_STREAM and dev in _side:
        torch.cuda.current_stream(dev).wait_stream(_side[dev])


def _prepack(model, blk, idx, transposes):
    """Build block ``idx``'s bf16 weight copies (and, for the backward, their transposes) on the side stream while the
    main stream computes the neighbouring block: after an optimizer step every copy is stale, and the ~20 small cast /
    transpose kernels per block (93 MB written) otherwise sit between the block's GEMMs on the main stream.  The
    caller's next use of the block is ordered behind them by _wgrad_join."""
    from . import model as _model_mod
    if not _WGRAD_STREAM or _model_mod._Packed.always_rebuild:     # (under hipGraph capture the packs are graph nodes)
        return
    dev = blk.modulation.device
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(main)
    sa, ca = blk.self_attn, blk.cross_attn
    frozen_ffn = getattr(model, "reference_ffn_freeze", True) and idx > 10
    with torch.cuda.stream(side), torch.no_grad():
        wqk, _ = sa._w_qk()
        packs = {"qk": (sa, wqk), "v": (sa, sa._w("v")[0]), "o": (sa, sa._w("o")[0]),
                 "q": (ca, ca._w("q")[0]), "k": (ca, ca._w("k")[0]), "v_": (ca, ca._w("v")[0]), "o_": (ca, ca._w("o")[0])}
        if hasattr(ca, "k_img"):
            packs["k_img"] = (ca, ca._w("k_img")[0])
            packs["v_img"] = (ca, ca._w("v_img")[0])
        w1, w2 = blk._ffn_w(0)[0], blk._ffn_w(2)[0]
        if transposes:
            for key, (mod, w) in packs.items():
                _wT(mod, key.rstrip("_"), w)
            if not frozen_ffn:
                _wT(blk, "ffn2", w2)
                _wT(blk, "ffn0", w1)


def _wgrad(dy, x, xT=None, out=None):
    """dW[N, K] = dy[R, N]^T @ x[R, K]  (fp32) on dy and x as they are (row-major bf16): the k-major GEMM of
    csrc/gemm_tn.hip — no transposed copies.  With ``out`` the product is ADDED to it (a weight used twice in the
    block).  ``xT`` is ignored (kept for the OMH_WGRAD=nt path: two transposes + the NT kernel, for A/B timing)."""
    if _WGRAD_TN and _WGRAD_STREAM and _side.get("on"):
        dev = dy.device
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        if out is None:
            out = torch.empty(dy.shape[1], x.shape[1], dtype=torch.float32, device=dev)
            acc = False
        else:
            acc = True
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.gemm_tn(dy, x, out=out, accumulate=acc)
        for t in (dy, x, out):
            t.record_stream(side)
        return out
    if _WGRAD_TN:
        return ops.gemm_tn(dy, x, out=out, accumulate=out is not None)
    dyT = ops.transpose_bf16(dy)
    xT = ops.transpose_bf16(x) if xT is None else xT
    N, K, Rp = dyT.shape[0], xT.shape[0], dyT.shape[1]
    acc = out is not None
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    ops.gemm_raw(ptr(dyT), ptr(xT), ptr(out), N, K, Rp, Rp, Rp, K, EPI_ACC if acc else EPI_F32)
    return out


def _dgrad_ctx(dy, wT, d_ctx, first, L):
    """d_ctx[b, first:first+L, :] += dy[b*L:(b+1)*L, :] @ W  for every sample b (the context gradient of a K / V
    projection that reads a slice of the context rows): one batched GEMM into the strided destination."""
    B, Lc, d = d_ctx.shape
    N = dy.shape[1]
    ops.gemm_raw(ptr(dy), ptr(wT), ptr(d_ctx, first * d), L, d, N, dy.stride(0), wT.stride(0), d, EPI_ACC, batch=B,
                 strideA=L * dy.stride(0), strideB=0, strideC=Lc * d, b_kmajor=_DGRAD_NN)


class _ZeroArena:
    """One zero-filled fp32 buffer per block backward, carved into the ~20 small accumulators (bias, norm-gain and
    modulation gradients) that the atomics-based kernels add into: one fill launch instead of twenty."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.pos = 0
        self.pending = []                                       # deferred column sums: (dy, out) pairs

    def flush(self):
        """All bias gradients of the block in one launch (each alone is a 10-15 us latency-bound kernel)."""
        if self.pending:
            ops.colsum_accum_multi(self.pending)
            self.pending = []

    def take(self, *shape):
        n = 1
        for s_ in shape:
            n *= s_
        n_al = (n + 63) // 64 * 64                              # keep every slice 256-byte aligned
        if self.pos + n_al > self.buf.numel():                  # undersized estimate: fall back to a fresh buffer
            return torch.zeros(*shape, dtype=torch.float32, device=self.buf.device)
        out = self.buf[self.pos:self.pos + n].view(*shape)
        self.pos += n_al
        return out


_BGRAD_MULTI = os.environ.get("OMH_BGRAD", "multi") != "single"      # "single": one launch per bias gradient (A/B timing)


def _bgrad(dy, arena=None):
    """Bias gradient = column sums of dy.  With an arena the sum is deferred to ``arena.flush()`` at the end of the
    block backward (dy stays referenced until then); the returned accumulator is complete after the flush."""
    out = arena.take(dy.shape[1]) if arena is not None else torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
    if arena is not None and _BGRAD_MULTI:
        arena.pending.append((dy, out))
        return out
    return ops.colsum_accum(dy, out)


def _dgrad(dy, wT, out=None, accumulate=False):
    """dx[R, K] = dy[R, N] @ W[N, K]  with wT from _wT (W itself, or W^T bf16 [K, Np]); fp32 output."""
    R, N = dy.shape
    K = wT.shape[1] if _DGRAD_NN else wT.shape[0]
    if out is None:
        out = torch.empty(R, K, dtype=torch.float32, device=dy.device)
    ops.gemm_raw(ptr(dy), ptr(wT), ptr(out), R, K, N, dy.stride(0), wT.stride(0), out.stride(0),
                 EPI_ACC if accumulate else EPI_F32, b_kmajor=_DGRAD_NN)
    return out


def _axpy_rows(dst2d, src2d):
    """dst[b] += src[b] for small fp32 [B, n] buffers (colsum kernel with one row)."""
    for b in range(dst2d.shape[0]):
        ops.colsum_accum(src2d[b:b + 1], dst2d[b])


# ----------------------------------------------------------------------------- attention (training)
def _vt_from_v(v, B, L, d):
    Lp = _ru(L, 64)
    vt = torch.empty(B, d, Lp, dtype=torch.bfloat16, device=v.device)
    if Lp != L:
        vt[:, :, L:].zero_()                       # pad columns only; the transpose writes the rest
    ops.transpose_bf16_raw(ptr(v), ptr(vt), L, d, d, Lp, batch=B, bs_in=L * d, bs_out=d * Lp)
    return vt, Lp


def _attn_fwd(q, k, v, klens32, B, Lq, Lk, H, D, want_lse=False):
    d = H * D
    vt, Lp = _vt_from_v(v, B, Lk, d)
    o = torch.empty(B * Lq, d, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device) if want_lse else None
    ops.flash_attn_raw(ptr(q), ptr(k), ptr(vt), ptr(o), ptr(klens32) if klens32 is not None else None, B, H, Lq, Lk,
                       Lq * d, d, Lk * d, d, d * Lp, Lq * d, d, Lp, D ** -0.5, lse=ptr(lse) if want_lse else None)
    return (o, lse) if want_lse else o


# OMH_ATTN_BWD=unfused keeps the first implementation (scores materialised per head through the GEMM kernel, ~14
# launches per sample) for A/B timing and as a second opinion in the tests; the default is the fused kernel pair
# of csrc/attention_bwd.hip (3 launches + 3 transposes per call, whole batch at once).
_FUSED_ATTN_BWD = os.environ.get("OMH_ATTN_BWD", "fused") != "unfused"
_WGRAD_TN = os.environ.get("OMH_WGRAD", "tn") != "nt"


def _attn_bwd(q, k, v, do, klens, B, Lq, Lk, H, D):
    """Un-fused attention backward.  q/do bf16 [B*Lq, d]; k/v bf16 [B*Lk, d]; klens python ints.
    Returns fp32 dq [B*Lq, d], dk, dv [B*Lk, d]."""
    d = H * D
    dev = q.device
    scale = D ** -0.5
    dq = torch.zeros(B * Lq, d, dtype=torch.float32, device=dev)
    dk = torch.zeros(B * Lk, d, dtype=torch.float32, device=dev)
    dv = torch.zeros(B * Lk, d, dtype=torch.float32, device=dev)
    Sp = _ru(Lq, 8)
    for b in range(B):
        L = min(int(klens[b]), Lk)
        if L <= 0:
            continue
        Lp = _ru(L, 8)
        qb, dob = q[b * Lq:(b + 1) * Lq], do[b * Lq:(b + 1) * Lq]
        kb, vb = k[b * Lk:b * Lk + L], v[b * Lk:b * Lk + L]
        sc = torch.empty(H, Lq, Lp, dtype=torch.float32, device=dev)
        ops.gemm_raw(ptr(qb), ptr(kb), ptr(sc), Lq, L, D, d, d, Lp, EPI_F32, batch=H, strideA=D, strideB=D,
                     strideC=Lq * Lp)
        p = torch.zeros(H * Lq, Lp, dtype=torch.bfloat16, device=dev)
        ops.softmax_rows(sc.view(H * Lq, Lp), p, L, scale)
        dp = sc                                                    # reuse the score buffer
        ops.gemm_raw(ptr(dob), ptr(vb), ptr(dp), Lq, L, D, d, d, Lp, EPI_F32, batch=H, strideA=D, strideB=D,
                     strideC=Lq * Lp)
        ds = torch.zeros(H * Lq, Lp, dtype=torch.bfloat16, device=dev)
        ops.softmax_bwd_rows(p, dp.view(H * Lq, Lp), ds, L, scale)
        # dq_h = dS_h K_h
        kT = ops.transpose_bf16(kb)                                # [d, Lp]
        ops.gemm_raw(ptr(ds), ptr(kT), ptr(dq, b * Lq * d), Lq, D, Lp, Lp, Lp, d, EPI_F32, batch=H,
                     strideA=Lq * Lp, strideB=D * Lp, strideC=D)
        # dk_h = dS_h^T Q_h ; dv_h = P_h^T dO_h
        dsT = torch.zeros(H, Lp, Sp, dtype=torch.bfloat16, device=dev)
        ops.transpose_bf16_raw(ptr(ds), ptr(dsT), Lq, Lp, Lp, Sp, batch=H, bs_in=Lq * Lp, bs_out=Lp * Sp)
        qT = ops.transpose_bf16(qb)                                # [d, Sp]
        ops.gemm_raw(ptr(dsT), ptr(qT), ptr(dk, b * Lk * d), L, D, Sp, Sp, Sp, d, EPI_F32, batch=H,
                     strideA=Lp * Sp, strideB=D * Sp, strideC=D)
        pT = dsT                                                   # reuse (same shape, fully rewritten below)
        ops.transpose_bf16_raw(ptr(p), ptr(pT), Lq, Lp, Lp, Sp, batch=H, bs_in=Lq * Lp, bs_out=Lp * Sp)
        doT = ops.transpose_bf16(dob)
        ops.gemm_raw(ptr(pT), ptr(doT), ptr(dv, b * Lk * d), L, D, Sp, Sp, Sp, d, EPI_F32, batch=H,
                     strideA=Lp * Sp, strideB=D * Sp, strideC=D)
    return dq, dk, dv


# ----------------------------------------------------------------------------- block node
class _BlockFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, model, st, idx, *params):
        block = model.blocks[idx]
        fc = st.fc
        _wgrad_join(x.device)                                     # this block's packs (prefetched by the previous one)
        if idx + 1 < len(model.blocks):
            _prepack(model, model.blocks[idx + 1], idx + 1, False)
        with torch.no_grad():
            x_out = x.detach().clone()
            seq_lens = fc.seq_lens32.long()
            out = block(x_out, fc.e0, seq_lens, fc.grid32.long(), (fc.rope_cos, fc.rope_sin), fc.ctx,
                        fc.ctx_lens32.long(), block_idx=idx, _fc=fc)
        ctx.save_for_backward(x.detach())
        ctx.model, ctx.st, ctx.idx = model, st, idx
        return out

    @staticmethod
    def backward(ctx, dx_out):
        (x0,) = ctx.saved_tensors
        model, st, idx = ctx.model, ctx.st, ctx.idx
        blk = model.blocks[idx]
        names = [n for n, _ in blk.named_parameters()]
        if idx > 0:
            _prepack(model, model.blocks[idx - 1], idx - 1, True)
        with torch.no_grad():
            grads = _block_backward(model, blk, idx, st, x0, dx_out.float().contiguous().clone())
        out = []
        for n, p in zip(names, blk.parameters()):
            g = grads.get(n) if p.requires_grad else None
            out.append(None if g is None else g.view(p.shape).to(p.dtype))
        return (grads["__dx__"], None, None, None, *out)


def _block_backward(model, blk, idx, st, x0, dx):
    """Recompute block ``idx`` from its input x0 (fp32 [B,S,d]) and back-propagate dx (in place)."""
    _side["on"] = True                                            # weight gradients of this block: second stream (if enabled)
    fc = st.fc
    B, S, d = x0.shape
    R = B * S
    dev = x0.device
    sa, ca = blk.self_attn, blk.cross_attn
    N, D = sa.num_heads, sa.head_dim
    eps = blk.eps
    mod = blk.modulation.detach().float().contiguous()
    e0 = fc.e0
    six = 6 * d
    i2v = hasattr(ca, "k_img")                                        # model.py:189-230: extra attention over the 257 image tokens
    n_img = 257 if i2v else 0
    frozen_ffn = getattr(model, "reference_ffn_freeze", True) and idx > 10
    seq_lens, ctx_lens = fc.seq_lens_host, fc.ctx_lens_host
    Lc = fc.Lc
    Lt = Lc - n_img                                                   # text tokens
    arena = _ZeroArena(B * six + 2 * six + 16 * d + 2 * blk.ffn[0].out_features + 4096, dev)
    d_eb = arena.take(B, 6, d)                                        # grads of e = modulation + e0
    g = {}

    def ln_fwd(xin, shift_i, scale_i):
        h = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
        ops.layernorm_modulate_raw(ptr(xin), ptr(h), R, d, eps, 1.0, ptr(mod, scale_i * d), ptr(e0, scale_i * d), six,
                                   ptr(mod, shift_i * d), ptr(e0, shift_i * d), six, S)
        return h

    def ln_bwd(xin, dh, shift_i, scale_i):
        ops.layernorm_modulate_bwd_raw(ptr(xin), ptr(dh), ptr(dx), R, d, eps, 1.0, ptr(mod, scale_i * d),
                                       ptr(e0, scale_i * d), six, ptr(d_eb, scale_i * d), ptr(d_eb, shift_i * d),
                                       six, S)

    def lin(a, w, b, epi=EPI_BF16):
        return ops.gemm(a, w, bias=b, epilogue=epi)

    def resid_fwd(xin, y, gate_i):
        xo = torch.empty_like(xin)
        if gate_i is None:
            ops.gated_residual_fwd_raw(ptr(xin), ptr(y), ptr(xo), R, d, 1.0, None, None, 0, S)
        else:
            ops.gated_residual_fwd_raw(ptr(xin), ptr(y), ptr(xo), R, d, 0.0, ptr(mod, gate_i * d),
                                       ptr(e0, gate_i * d), six, S)
        return xo

    def resid_bwd(y, gate_i):
        dy = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
        if gate_i is None:
            ops.gated_residual_bwd_raw(ptr(dx), None, ptr(dy), None, 0, R, d, 1.0, None, None, 0, S)
        else:
            ops.gated_residual_bwd_raw(ptr(dx), ptr(y), ptr(dy), ptr(d_eb, gate_i * d), six, R, d, 0.0,
                                       ptr(mod, gate_i * d), ptr(e0, gate_i * d), six, S)
        return dy

    # ================= recompute (un-fused, keeps intermediates) =================
    x0 = x0.contiguous()
    h1 = ln_fwd(x0, 0, 1)
    wqk, bqk = sa._w_qk()
    qk_pre = lin(h1, wqk, bqk, EPI_F32)                                   # [R, 2d] fp32
    q = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
    k = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
    nq, nk = sa._norm_w("norm_q"), sa._norm_w("norm_k")
    for dst, off, w in ((q, 0, nq), (k, d, nk)):
        ops.rmsnorm_rope_raw(ptr(qk_pre, off), 2 * d, ptr(dst), R, d, ptr(w) if w is not None else None, sa.eps,
                             int(sa.qk_norm), ptr(fc.rope_cos), ptr(fc.rope_sin), fc.rope_cos.shape[0], D,
                             ptr(fc.grid32), S)
    wv, bv = sa._w("v")
    v = lin(h1, wv, bv)
    o, lse_sa = _attn_fwd(q, k, v, fc.seq_lens32, B, S, S, N, D, want_lse=True)
    wo, bo = sa._w("o")
    y1 = lin(o, wo, bo)
    x1 = resid_fwd(x0, y1, 2)
    # cross attention
    h3 = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
    if blk.cross_attn_norm:
        n3 = blk.norm3
        w3, b3 = n3.weight.detach().float().contiguous(), n3.bias.detach().float().contiguous()
        ops.layernorm_modulate_raw(ptr(x1), ptr(h3), R, d, n3.eps, 0.0, ptr(w3), None, 0, ptr(b3), None, 0, R)
    else:
        ops.cast_bf16(x1.view(R, d), out=h3)
    wqc, bqc = ca._w("q")
    qc_pre = lin(h3, wqc, bqc, EPI_F32)
    qc = ops.rmsnorm_rope(qc_pre, ca._norm_w("norm_q"), ca.eps, do_norm=ca.qk_norm)
    ctx2 = fc.ctx.view(B * Lc, d) if not i2v else fc.ctx[:, n_img:].contiguous().view(B * Lt, d)
    wkc, bkc = ca._w("k")
    wvc, bvc = ca._w("v")
    kc_pre = lin(ctx2, wkc, bkc, EPI_F32)
    kc = ops.rmsnorm_rope(kc_pre, ca._norm_w("norm_k"), ca.eps, do_norm=ca.qk_norm)
    vc = lin(ctx2, wvc, bvc)
    oc, lse_ca = _attn_fwd(qc, kc, vc, fc.ctx_lens32, B, S, Lt, N, D, want_lse=True)
    woc, boc = ca._w("o")
    if i2v:
        ctxi = fc.ctx[:, :n_img].contiguous().view(B * n_img, d)
        wki, bki = ca._w("k_img")
        wvi, bvi = ca._w("v_img")
        ki_pre = lin(ctxi, wki, bki, EPI_F32)
        ki = ops.rmsnorm_rope(ki_pre, ca._norm_w("norm_k_img"), ca.eps, do_norm=ca.qk_norm)
        vi = lin(ctxi, wvi, bvi)
        oi, lse_ci = _attn_fwd(qc, ki, vi, None, B, S, n_img, N, D, want_lse=True)
        y2f = lin(oc, woc, boc, EPI_F32)                              # o(o_text + o_img): two products into one fp32 sum
        ops.gemm(oi, woc, out=y2f, epilogue=EPI_ACC)
        y2 = ops.cast_bf16(y2f)
    else:
        y2 = lin(oc, woc, boc)
    x2 = resid_fwd(x1, y2, None)
    # FFN
    h2 = ln_fwd(x2, 3, 4)
    w1, b1 = blk._ffn_w(0)
    w2, b2 = blk._ffn_w(2)
    u_pre = lin(h2, w1, b1)
    u = ops.gelu_tanh(u_pre)
    y3 = lin(u, w2, b2)

    # ================= backward =================
    # ---- FFN branch: x3 = x2 + y3 * g5
    dy3 = resid_bwd(y3, 5)
    if not frozen_ffn:
        g["ffn.2.weight"], g["ffn.2.bias"] = _wgrad(dy3, u), _bgrad(dy3, arena)
        du = ops.gemm(dy3, _wT(blk, "ffn2", w2), epilogue=EPI_BF16, b_kmajor=_DGRAD_NN)       # [R, ffn]
        du_pre = ops.gelu_tanh_bwd(du, u_pre)
        g["ffn.0.weight"], g["ffn.0.bias"] = _wgrad(du_pre, h2), _bgrad(du_pre, arena)
        dh2 = _dgrad(du_pre, _wT(blk, "ffn0", w1))
        ln_bwd(x2, dh2, 3, 4)
        del du, du_pre, dh2
    del dy3, u, u_pre, y3, h2
    # ---- cross-attention branch: x2 = x1 + y2
    dy2 = resid_bwd(None, None)
    g["cross_attn.o.weight"], g["cross_attn.o.bias"] = _wgrad(dy2, oc), _bgrad(dy2, arena)
    doc = ops.gemm(dy2, _wT(ca, "o", woc), epilogue=EPI_BF16, b_kmajor=_DGRAD_NN)
    if _FUSED_ATTN_BWD:
        dqc, dkc, dvc = ops.flash_attn_bwd(qc, kc, vc, oc, doc, lse_ca, fc.ctx_lens32, B, N, S, Lt, D ** -0.5)
    else:
        dqc, dkc, dvc = _attn_bwd(qc, kc, vc, doc, ctx_lens, B, S, Lt, N, D)
    if i2v:                                                           # the image-token branch: same q, same dO
        _wgrad(dy2, oi, out=g["cross_attn.o.weight"])
        dqi, dki, dvi = ops.flash_attn_bwd(qc, ki, vi, oi, doc, lse_ci, None, B, N, S, n_img, D ** -0.5)
        ops.colsum_accum(dqi.view(1, R * d), dqc.view(R * d))         # dq = dq_text + dq_img
        Ri = B * n_img
        dki_pre = torch.empty(Ri, d, dtype=torch.bfloat16, device=dev)
        dnki = arena.take(d) if ca.qk_norm else None
        ops.rmsnorm_rope_bwd_raw(ptr(ki_pre), d, ptr(dki), d, ptr(dki_pre), d, ptr(dnki) if dnki is not None else None,
                                 Ri, d, ptr(ca._norm_w("norm_k_img")) if ca.qk_norm else None, ca.eps, int(ca.qk_norm),
                                 None, None, 0, D, None, 0)
        if dnki is not None:
            g["cross_attn.norm_k_img.weight"] = dnki
        ctxiT = None if _WGRAD_TN else ops.transpose_bf16(ctxi)
        g["cross_attn.k_img.weight"], g["cross_attn.k_img.bias"] = _wgrad(dki_pre, ctxi, ctxiT), _bgrad(dki_pre, arena)
        dvi_b = ops.cast_bf16(dvi)
        g["cross_attn.v_img.weight"], g["cross_attn.v_img.bias"] = _wgrad(dvi_b, ctxi, ctxiT), _bgrad(dvi_b, arena)
        _dgrad_ctx(dki_pre, _wT(ca, "k_img", wki), st.d_ctx, 0, n_img)
        _dgrad_ctx(dvi_b, _wT(ca, "v_img", wvi), st.d_ctx, 0, n_img)
        del dqi, dki, dvi, dki_pre, dvi_b, ki, vi, oi, ki_pre, ctxi, ctxiT
    dqc_pre = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
    dnq = arena.take(d) if ca.qk_norm else None
    ops.rmsnorm_rope_bwd_raw(ptr(qc_pre), d, ptr(dqc), d, ptr(dqc_pre), d, ptr(dnq) if dnq is not None else None, R, d,
                             ptr(ca._norm_w("norm_q")) if ca.qk_norm else None, ca.eps, int(ca.qk_norm), None, None,
                             0, D, None, 0)
    if dnq is not None:
        g["cross_attn.norm_q.weight"] = dnq
    g["cross_attn.q.weight"], g["cross_attn.q.bias"] = _wgrad(dqc_pre, h3), _bgrad(dqc_pre, arena)
    dh3 = _dgrad(dqc_pre, _wT(ca, "q", wqc))
    Rc = B * Lt
    dkc_pre = torch.empty(Rc, d, dtype=torch.bfloat16, device=dev)
    dnk = arena.take(d) if ca.qk_norm else None
    ops.rmsnorm_rope_bwd_raw(ptr(kc_pre), d, ptr(dkc), d, ptr(dkc_pre), d, ptr(dnk) if dnk is not None else None, Rc, d,
                             ptr(ca._norm_w("norm_k")) if ca.qk_norm else None, ca.eps, int(ca.qk_norm), None, None,
                             0, D, None, 0)
    if dnk is not None:
        g["cross_attn.norm_k.weight"] = dnk
    ctx2T = None if _WGRAD_TN else ops.transpose_bf16(ctx2)
    g["cross_attn.k.weight"], g["cross_attn.k.bias"] = _wgrad(dkc_pre, ctx2, ctx2T), _bgrad(dkc_pre, arena)
    dvc_b = ops.cast_bf16(dvc)
    g["cross_attn.v.weight"], g["cross_attn.v.bias"] = _wgrad(dvc_b, ctx2, ctx2T), _bgrad(dvc_b, arena)
    _dgrad_ctx(dkc_pre, _wT(ca, "k", wkc), st.d_ctx, n_img, Lt)
    _dgrad_ctx(dvc_b, _wT(ca, "v", wvc), st.d_ctx, n_img, Lt)
    if blk.cross_attn_norm:
        dw3, db3 = arena.take(d), arena.take(d)
        ops.layernorm_modulate_bwd_raw(ptr(x1), ptr(dh3), ptr(dx), R, d, blk.norm3.eps, 0.0, ptr(w3), None, 0,
                                       ptr(dw3), ptr(db3), 0, R)
        g["norm3.weight"], g["norm3.bias"] = dw3, db3
    else:
        ops.colsum_accum(dh3.view(1, R * d), dx.view(R * d))                  # dx += dh3
    del dy2, doc, dqc, dkc, dvc, dqc_pre, dkc_pre, dvc_b, dh3, qc, kc, vc, oc, qc_pre, kc_pre, y2, h3
    # ---- self-attention branch: x1 = x0 + y1 * g2
    dy1 = resid_bwd(y1, 2)
    g["self_attn.o.weight"], g["self_attn.o.bias"] = _wgrad(dy1, o), _bgrad(dy1, arena)
    do = ops.gemm(dy1, _wT(sa, "o", wo), epilogue=EPI_BF16, b_kmajor=_DGRAD_NN)
    if _FUSED_ATTN_BWD:
        dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, do, lse_sa, fc.seq_lens32, B, N, S, S, D ** -0.5)
    else:
        dq, dk, dv = _attn_bwd(q, k, v, do, seq_lens, B, S, S, N, D)
    dqk_pre = torch.empty(R, 2 * d, dtype=torch.bfloat16, device=dev)
    for off, dyy, w, nm in ((0, dq, nq, "norm_q"), (d, dk, nk, "norm_k")):
        dnw = arena.take(d) if sa.qk_norm else None
        ops.rmsnorm_rope_bwd_raw(ptr(qk_pre, off), 2 * d, ptr(dyy), d, ptr(dqk_pre, off), 2 * d,
                                 ptr(dnw) if dnw is not None else None, R, d, ptr(w) if w is not None else None,
                                 sa.eps, int(sa.qk_norm), ptr(fc.rope_cos), ptr(fc.rope_sin), fc.rope_cos.shape[0],
                                 D, ptr(fc.grid32), S)
        if dnw is not None:
            g[f"self_attn.{nm}.weight"] = dnw
    h1T = None if _WGRAD_TN else ops.transpose_bf16(h1)
    dwqk, dbqk = _wgrad(dqk_pre, h1, h1T), _bgrad(dqk_pre, arena)
    g["self_attn.q.weight"], g["self_attn.k.weight"] = dwqk[:d], dwqk[d:]
    g["self_attn.q.bias"], g["self_attn.k.bias"] = dbqk[:d], dbqk[d:]
    dv_b = ops.cast_bf16(dv)
    g["self_attn.v.weight"], g["self_attn.v.bias"] = _wgrad(dv_b, h1, h1T), _bgrad(dv_b, arena)
    dh1 = _dgrad(dqk_pre, _wT(sa, "qk", wqk))
    _dgrad(dv_b, _wT(sa, "v", wv), out=dh1, accumulate=True)
    ln_bwd(x0, dh1, 0, 1)
    # ---- modulation / e0
    dmod = arena.take(six)
    ops.colsum_accum(d_eb.view(B, six), dmod)
    g["modulation"] = dmod
    _axpy_rows(st.d_e0.view(B, six), d_eb.view(B, six))
    arena.flush()
    _side["on"] = False
    _wgrad_join(dev)
    g["__dx__"] = dx
    return g


# ----------------------------------------------------------------------------- head node
class _HeadFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, model, st, grids, *params):
        with torch.no_grad():
            out = model.head(x.detach(), st.e)
            outs = model.unpatchify(out, None, _grids=grids)
        ctx.save_for_backward(x.detach())
        ctx.model, ctx.st, ctx.grids = model, st, grids
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        (x,) = ctx.saved_tensors
        model, st, grids = ctx.model, ctx.st, ctx.grids
        head = model.head
        B, S, d = x.shape
        R = B * S
        dev = x.device
        with torch.no_grad():
            mod = head.modulation.detach().float().contiguous()
            e = st.e.contiguous()
            ncol = head.head.weight.shape[0]
            dtok = torch.zeros(R, ncol, dtype=torch.bfloat16, device=dev)
            for b, (gb, grd) in enumerate(zip(gouts, grids)):
                if gb is None:
                    continue
                n = grd[0] * grd[1] * grd[2]
                dtok[b * S:b * S + n].copy_(ops.unpatchify_bwd(gb.contiguous().float(), grd, model.patch_size))
            hh = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
            ops.layernorm_modulate_raw(ptr(x), ptr(hh), R, d, head.eps, 1.0, ptr(mod, d), ptr(e), d, ptr(mod, 0),
                                       ptr(e), d, S)
            w, _ = head._packed.get("head", (head.head.weight, head.head.bias), lambda: (
                ops.cast_bf16(head.head.weight.detach().float().contiguous()),
                head.head.bias.detach().float().contiguous()))
            gw, gb_ = _wgrad(dtok, hh), _bgrad(dtok)
            dhh = _dgrad(dtok, _wT(head, "head", w))
            dx = torch.zeros(R, d, dtype=torch.float32, device=dev)
            d_hb = torch.zeros(B, 2, d, dtype=torch.float32, device=dev)
            ops.layernorm_modulate_bwd_raw(ptr(x), ptr(dhh), ptr(dx), R, d, head.eps, 1.0, ptr(mod, d), ptr(e), d,
                                           ptr(d_hb, d), ptr(d_hb, 0), 2 * d, S)
            dmod = torch.zeros(2 * d, dtype=torch.float32, device=dev)
            ops.colsum_accum(d_hb.view(B, 2 * d), dmod)
            for b in range(B):                       # e enters both the scale and the shift (model.py:357-358)
                ops.colsum_accum(d_hb[b], st.d_e[b])
        grads = {"modulation": dmod.view(1, 2, d), "head.weight": gw, "head.bias": gb_}
        out = []
        for n, p in head.named_parameters():
            gg = grads.get(n) if p.requires_grad else None
            out.append(None if gg is None else gg.view(p.shape).to(p.dtype))
        return (dx.view(B, S, d), None, None, None, *out)


# ----------------------------------------------------------------------------- embed node
_EMBED_MODULES = ("patch_embedding", "text_embedding", "time_embedding", "time_projection")


def _embed_params(model):
    out = []
    for m in _EMBED_MODULES + (("img_emb",) if hasattr(model, "img_emb") else ()):
        out += [(f"{m}.{n}", p) for n, p in getattr(model, m).named_parameters()]
    return out


def _img_emb_backward(model, clip_fea, d_img, g):
    """Backward of MLPProj (model.py:362-374: LayerNorm, Linear, GELU(erf), Linear, LayerNorm) on the CLIP tokens:
    recompute keeping the pre-activations, then the chain rule.  d_img fp32 [B*257, dim] = gradient of its output."""
    proj = model.img_emb.proj
    ln0, l1, l3, ln4 = proj[0], proj[1], proj[3], proj[4]
    dev = d_img.device
    x = clip_fea.to(device=dev, dtype=torch.float32).contiguous().view(-1, clip_fea.shape[-1])
    rows, cin = x.shape
    dim = l3.weight.shape[0]
    f = lambda t_: t_.detach().float().contiguous()
    w0, b0, w4, b4 = f(ln0.weight), f(ln0.bias), f(ln4.weight), f(ln4.bias)
    h0 = ops.layernorm_modulate(x, ln0.eps, 0.0, mul0=w0, add0=b0)
    w1, w3 = ops.cast_bf16(f(l1.weight)), ops.cast_bf16(f(l3.weight))
    z1 = ops.gemm(h0, w1, bias=f(l1.bias), epilogue=EPI_BF16)
    g1 = ops.gelu_erf(z1)
    z3 = ops.gemm(g1, w3, bias=f(l3.bias), epilogue=EPI_F32)
    # LayerNorm 4 (affine): dz3, dw4, db4
    dz3 = torch.zeros(rows, dim, dtype=torch.float32, device=dev)
    dw4, db4 = torch.zeros(dim, dtype=torch.float32, device=dev), torch.zeros(dim, dtype=torch.float32, device=dev)
    ops.layernorm_modulate_bwd_raw(ptr(z3), ptr(d_img), ptr(dz3), rows, dim, ln4.eps, 0.0, ptr(w4), None, 0, ptr(dw4),
                                   ptr(db4), 0, rows)
    dz3b = ops.cast_bf16(dz3)
    g["img_emb.proj.4.weight"], g["img_emb.proj.4.bias"] = dw4, db4
    g["img_emb.proj.3.weight"], g["img_emb.proj.3.bias"] = _wgrad(dz3b, g1), _bgrad(dz3b)
    dg1 = ops.gemm(dz3b, _wT_once(w3), epilogue=EPI_BF16, b_kmajor=_DGRAD_NN)
    dz1 = ops.gelu_erf_bwd(dg1, z1)
    g["img_emb.proj.1.weight"], g["img_emb.proj.1.bias"] = _wgrad(dz1, h0), _bgrad(dz1)
    dh0 = _dgrad(dz1, _wT_once(w1))
    dx = torch.zeros(rows, cin, dtype=torch.float32, device=dev)      # (gradient w.r.t. the CLIP tokens: not needed)
    dw0, db0 = torch.zeros(cin, dtype=torch.float32, device=dev), torch.zeros(cin, dtype=torch.float32, device=dev)
    ops.layernorm_modulate_bwd_raw(ptr(x), ptr(dh0), ptr(dx), rows, cin, ln0.eps, 0.0, ptr(w0), None, 0, ptr(dw0),
                                   ptr(db0), 0, rows)
    g["img_emb.proj.0.weight"], g["img_emb.proj.0.bias"] = dw0, db0


class _EmbedFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, st, x_list, t, context, seq_len, clip_fea, y, extra_tokens, *params):
        with torch.no_grad():
            xs, e, fc, grids, lens, ctx_lens = model._embed(x_list, t, context, seq_len, clip_fea, y, extra_tokens)
        ctx.n_extra = 0 if extra_tokens is None else int(extra_tokens.shape[1])
        ctx.extra_dtype = None if extra_tokens is None else extra_tokens.dtype
        st.fc, st.e, st.grids, st.lens = fc, e, grids, lens
        B, d = fc.B, fc.dim
        dev = xs.device
        st.d_e0 = torch.zeros(B, 6, d, dtype=torch.float32, device=dev)
        st.d_e = torch.zeros(B, d, dtype=torch.float32, device=dev)
        st.d_ctx = torch.zeros(B, fc.Lc, d, dtype=torch.float32, device=dev)
        ctx.model, ctx.st = model, st
        ctx.inputs = (x_list, t, context, seq_len, y, clip_fea)
        return xs

    @staticmethod
    def backward(ctx, dxs):
        model, st = ctx.model, ctx.st
        x_list, t, context, seq_len, y, clip_fea = ctx.inputs
        fc = st.fc
        B, d = fc.B, fc.dim
        dev = dxs.device
        g = {}
        with torch.no_grad():
            dxs = dxs.contiguous().float()
            # ---- patch embedding: x[b,:n] = tok_b Wpe^T + bpe
            pt, ph, pw = model.patch_size
            kin = model.in_dim * pt * ph * pw
            Kp = _ru(kin, 8)
            if y is not None:
                x_list = [torch.cat([u, v], dim=0) for u, v in zip(x_list, y)]
            dW = torch.zeros(d, Kp, dtype=torch.float32, device=dev)
            db = torch.zeros(d, dtype=torch.float32, device=dev)
            for b, u in enumerate(x_list):
                n = st.lens[b]
                tok = ops.patchify(u.to(device=dev, dtype=torch.float32).contiguous(), model.patch_size, Kp)
                dxb = ops.cast_bf16(dxs[b, :n].contiguous())
                if _WGRAD_TN:
                    ops.gemm_tn(dxb, tok, out=dW, accumulate=True)
                else:
                    dxT, tokT = ops.transpose_bf16(dxb), ops.transpose_bf16(tok)
                    ops.gemm_raw(ptr(dxT), ptr(tokT), ptr(dW), d, Kp, dxT.shape[1], dxT.shape[1], tokT.shape[1], Kp,
                                 EPI_ACC)
                ops.colsum_accum(dxs[b, :n], db)
            g["patch_embedding.weight"] = dW[:, :kin].contiguous()
            g["patch_embedding.bias"] = db
            # ---- time embedding (fp32): z0 = W0 sin + b0 ; e = W2 silu(z0) + b2 ; e0 = Wp silu(e) + bp
            te0, te2, tp1 = model.time_embedding[0], model.time_embedding[2], model.time_projection[1]
            W0, W2, Wp = (m.weight.detach().float().contiguous() for m in (te0, te2, tp1))
            sin = ops.sinusoidal_embedding(t.to(dev), model.freq_dim)
            z0 = ops.dense_f32(sin, W0, te0.bias.detach().float(), 0, 0)
            e = st.e
            dWp, dbp = torch.zeros_like(Wp), torch.zeros(Wp.shape[0], dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(e, Wp, st.d_e0.view(B, 6 * d), dW=dWp, db=dbp, dx=st.d_e, dx_accumulate=True, act_in=1)
            dW2, db2 = torch.zeros_like(W2), torch.zeros(d, dtype=torch.float32, device=dev)
            dz0 = torch.empty(B, d, dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(z0, W2, st.d_e, dW=dW2, db=db2, dx=dz0, act_in=1)
            dW0, db0 = torch.zeros_like(W0), torch.zeros(d, dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(sin, W0, dz0, dW=dW0, db=db0, dx=None, act_in=0)
            g.update({"time_projection.1.weight": dWp, "time_projection.1.bias": dbp, "time_embedding.2.weight": dW2,
                      "time_embedding.2.bias": db2, "time_embedding.0.weight": dW0, "time_embedding.0.bias": db0})
            # ---- text embedding: ctx = W2t gelu(W0t cin + b0t) + b2t
            ctx_in = torch.zeros(B, model.text_len, model.text_dim, dtype=torch.float32, device=dev)
            for b, u in enumerate(context):
                ctx_in[b, :u.shape[0]] = u.to(device=dev, dtype=torch.float32)
            t0, t2 = model.text_embedding[0], model.text_embedding[2]
            w0 = ops.cast_bf16(t0.weight.detach().float().contiguous())
            w2 = ops.cast_bf16(t2.weight.detach().float().contiguous())
            cin = ops.cast_bf16(ctx_in).view(B * model.text_len, model.text_dim)
            pre = ops.gemm(cin, w0, bias=t0.bias.detach().float(), epilogue=EPI_BF16)
            gl = ops.gelu_tanh(pre)
            n_img = fc.Lc - model.text_len
            dctx = st.d_ctx[:, n_img:].contiguous().view(B * model.text_len, d)
            dctx_b = ops.cast_bf16(dctx)
            g["text_embedding.2.weight"], g["text_embedding.2.bias"] = _wgrad(dctx_b, gl), _bgrad(dctx_b)
            dgl = ops.gemm(dctx_b, _wT_once(w2), epilogue=EPI_BF16, b_kmajor=_DGRAD_NN)
            dpre = ops.gelu_tanh_bwd(dgl, pre)
            g["text_embedding.0.weight"], g["text_embedding.0.bias"] = _wgrad(dpre, cin), _bgrad(dpre)
            # ---- image embedding (i2v): the first n_img context rows came from img_emb(clip_fea)
            if n_img and clip_fea is not None:
                _img_emb_backward(model, clip_fea, st.d_ctx[:, :n_img].contiguous().view(B * n_img, d), g)
        out = []
        for n, p in _embed_params(model):
            gg = g.get(n) if p.requires_grad else None
            out.append(None if gg is None else gg.view(p.shape).to(p.dtype))
        # condition tokens of the OmniHuman adapters (omnihuman_wan_t2v.py:453-488): they sit in front of the text
        # in the context, so their gradient is the leading rows of the context gradient
        d_extra = None
        if ctx.n_extra and ctx.needs_input_grad[8]:
            d_extra = st.d_ctx[:, :ctx.n_extra].contiguous().to(ctx.extra_dtype)
        return (None, None, None, None, None, None, None, None, d_extra, *out)


def forward_train(model, x, t, context, seq_len, clip_fea=None, y=None, extra_conditions=None):
    """WanModel.forward with autograd enabled: same outputs as the inference path, attached to a
    graph of hand-written nodes (see module docstring).  ``extra_conditions``: [B, Ne, dim] condition tokens (or a
    dict holding them under 'tokens') that receive a gradient like any other input."""
    st = _State()
    x_list = list(x) if not isinstance(x, (list, tuple)) else list(x)
    eparams = [p for _, p in _embed_params(model)]
    tok = extra_conditions.get("tokens") if isinstance(extra_conditions, dict) else extra_conditions
    xs = _EmbedFn.apply(model, st, x_list, t, list(context), seq_len, clip_fea, y, tok, *eparams)
    if not xs.requires_grad:
        xs.requires_grad_(True)          # keeps the chain alive when the embed parameters are frozen
    for i, blk in enumerate(model.blocks):
        xs = _BlockFn.apply(xs, model, st, i, *list(blk.parameters()))
    outs = _HeadFn.apply(xs, model, st, st.grids, *list(model.head.parameters()))
    return list(outs)
