"""Wan2.1 DiT backbone on hand-written gfx950 kernels — drop-in for the
reference's ``wan.modules.model`` (seaweed_apt/wan/modules/model.py).

Same constructor, attributes, module tree and state-dict keys as the
reference ``WanModel`` (model.py:377-499) so checkpoints, EMA
zips, forward hooks on ``model.blocks[i]`` and ``copy.deepcopy`` keep working;
``forward(x, t, context, seq_len, clip_fea=None, y=None)`` returns the same
list of fp32 ``[C_out, F, H, W]`` tensors.  The parameters stay ordinary fp32
``nn.Parameter``s; the arithmetic runs on bf16 copies packed once per weight
version, through the C ABI of libomh.so (include/omh.h):

  patch embed  -> omh_patchify + omh_gemm_bf16(EPI_F32)            model.py:515-522
  time embed   -> omh_sinusoidal_embedding + omh_dense_f32          model.py:526-528
  text embed   -> omh_cast + omh_gemm(GELU) + omh_gemm              model.py:531-532
  block        -> omh_layernorm_modulate, omh_gemm (q|k fp32, V^T bf16),
                  omh_rmsnorm_rope, omh_flash_attn_fwd_d128,
                  omh_gemm(EPI_RESID: x += (o W^T + b) * gate), ... model.py:279-330
  head         -> omh_layernorm_modulate + omh_gemm + omh_unpatchify model.py:349-359,565-588

Deliberate differences from the reference: the
reference's per-call ``empty_cache()``, FFN->CPU offload of blocks > 10 and
forced fp16 autocast are execution mechanics, not math, and are not
reproduced; compute is bf16 MFMA with fp32 accumulation, fp32 residual
stream, fp32 norm statistics (tolerance stated in tests/ and DESIGN.md).
"""
import math
from typing import List, Optional

import torch
import torch.nn as nn

from .._backend import ops

BIAS_M, BIAS_N, BIAS_NONE = ops.BIAS_M, ops.BIAS_N, ops.BIAS_NONE
EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID = ops.EPI_BF16, ops.EPI_F32, ops.EPI_GELU_BF16, ops.EPI_RESID
EPI_BF16_SPLIT_T = ops.EPI_BF16_SPLIT_T
ptr = ops.ptr

__all__ = ["WanModel"]


# ----------------------------------------------------------------------------
# reference-compatible helpers
# ----------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim, position):
    """model.py:17-27 (host fp64 version, kept for API parity)."""
    assert dim % 2 == 0
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_params(max_seq_len, dim, theta=10000):
    """model.py:31-38 — complex128 rotary table."""
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


class _Packed:
    """bf16 copies of weights, rebuilt when the source parameter changes."""

    def __init__(self):
        self.store = {}

    def get(self, key, params, builder):
        sig = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        ent = self.store.get(key)
        if ent is None or ent[0] != sig:
            with torch.no_grad():
                ent = (sig, builder())
            self.store[key] = ent
        return ent[1]

    def __deepcopy__(self, memo):
        return _Packed()


def _bf16(w: torch.Tensor) -> torch.Tensor:
    w = w.detach()
    if w.dtype == torch.bfloat16:
        return w.contiguous()
    return ops.cast_bf16(w.float().contiguous())


def _round_up(a, b):
    return (a + b - 1) // b * b


_CONST_CACHE = {}


def _dev_ints(values, dtype, device):
    """Small integer table (sequence lengths, grids) as a device tensor, cached by value: the forward issues no
    host-to-device copy after the first call with a given geometry (each one is a synchronous pageable copy, and
    none is allowed while a hipGraph is being captured).  Read-only by convention."""
    key = (str(device), dtype, tuple(tuple(v) if isinstance(v, (tuple, list)) else v for v in values))
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 4096:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


class _FwdCtx:
    """Per-forward shared state handed to every block."""
    __slots__ = ("B", "S", "dim", "e0", "seq_lens32", "ctx_lens32", "grid32", "rope_cos", "rope_sin", "ctx",
                 "Lc", "n_img", "seq_lens_host", "ctx_lens_host", "kv", "split_k")


class ContextState:
    """Everything a forward derives from (context, clip_fea) alone — the text/image embedding output and every
    block's cross-attention K (normalised) and V^T — computed once by ``WanModel.encode_context`` and accepted by
    ``WanModel.forward`` in place of ``context``.  The reference recomputes these on each of the 100 forwards
    of a 50-step CFG sample (model.py:531-537,176-178,216-220); they do not depend on x or t, so reusing them
    changes no value."""
    __slots__ = ("ctx", "ctx_lens", "kv", "B", "model_id", "version")


# ----------------------------------------------------------------------------
# modules (same tree / parameter names as the reference)
# ----------------------------------------------------------------------------
class WanRMSNorm(nn.Module):
    """model.py:72-88 (parameters only; fused into omh_rmsnorm_rope)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        return ops.rmsnorm_rope(x2, self.weight.detach(), self.eps).reshape(x.shape)


class WanLayerNorm(nn.LayerNorm):
    """model.py:91-104 (fused into omh_layernorm_modulate)."""

    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        x2 = x.float().contiguous()
        w = self.weight.detach() if self.elementwise_affine else None
        b = self.bias.detach() if self.elementwise_affine else None
        return ops.layernorm_modulate(x2, self.eps, 0.0 if w is not None else 1.0, mul0=w, add0=b)


class WanSelfAttention(nn.Module):
    """model.py:107-161."""

    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.o = nn.Linear(dim, dim)
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self._packed = _Packed()

    # packed weights ---------------------------------------------------------
    def _w_qk(self):
        return self._packed.get("qk", (self.q.weight, self.k.weight, self.q.bias, self.k.bias), lambda: (
            _bf16(torch.cat([self.q.weight.detach(), self.k.weight.detach()], 0)),
            torch.cat([self.q.bias.detach(), self.k.bias.detach()], 0).float().contiguous()))

    def _w_qkv(self):
        return self._packed.get("qkv", (self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias),
                                lambda: (_bf16(torch.cat([self.q.weight.detach(), self.k.weight.detach(),
                                                          self.v.weight.detach()], 0)),
                                         torch.cat([self.q.bias.detach(), self.k.bias.detach(), self.v.bias.detach()],
                                                   0).float().contiguous()))

    def _w(self, name):
        lin = getattr(self, name)
        return self._packed.get(name, (lin.weight, lin.bias),
                                lambda: (_bf16(lin.weight), lin.bias.detach().float().contiguous()))

    def _norm_w(self, name):
        m = getattr(self, name)
        return m.weight.detach() if isinstance(m, WanRMSNorm) else None

    def _attend(self, h, fc: "_FwdCtx"):
        """h bf16 [B*S, dim] -> attention output bf16 [B*S, dim] (before o-proj)."""
        B, S, d, N, D = fc.B, fc.S, self.dim, self.num_heads, self.head_dim
        R = B * S
        # q|k projection kept in bf16 (fp32 accumulate): the normalisation statistics are taken in fp32 from
        # it; measured effect on the 30-layer output < 1e-3 relative RMS, and it halves this step's traffic
        qk = torch.empty(R, 2 * d, dtype=torch.bfloat16, device=h.device)
        Sp = _round_up(S, 64)
        vt = torch.empty(B, d, Sp, dtype=torch.bfloat16, device=h.device)
        if Sp != S:
            vt[:, :, S:].zero_()                       # pad columns only (0 x P = 0 needs them finite); the GEMM writes the rest
        # long sequences: q | k | v as ONE product per clip over the concatenated weights (ABI v10) — h is read once,
        # the V third leaves the kernel transposed (V^T [dim, Sp], what the attention kernel reads); the same bits as the
        # two products below, which short sequences / batches keep (their rows fill the chip only together)
        # (tried at S = 1 560 in round 5: one launch instead of two is the same 19.7 ms per single-frame forward, and the
        # batched CFG pair gets slower, 14.8 -> 15.4 ms: one launch per clip against two over both clips)
        # (the library takes the one-launch form only when the q|k part is whole 384-column tiles: true at dim 1 536, not
        # at dim 5 120, where asking for it would cost B launches and a third weight copy for nothing — ADVICE round 5)
        fused = S >= 8192 and S % 8 == 0 and (2 * d) % 384 == 0
        if fused:
            wqkv, bqkv = self._w_qkv()
            for b in range(B):                         # one launch per clip: V^T is [B, dim, Sp], a clip's rows fill the chip
                ops.gemm_raw(ptr(h, b * S * d), ptr(wqkv), ptr(qk, b * S * 2 * d), S, 3 * d, d, d, d, 2 * d,
                             EPI_BF16_SPLIT_T, bias=ptr(bqkv), bias_mode=BIAS_N, aux=ptr(vt, b * d * Sp), ldaux=Sp,
                             n_split=2 * d)
        else:
            wqk, bqk = self._w_qk()
            ops.gemm_raw(ptr(h), ptr(wqk), ptr(qk), R, 2 * d, d, d, d, 2 * d, EPI_BF16, bias=ptr(bqk), bias_mode=BIAS_N)
        q = torch.empty(R, d, dtype=torch.bfloat16, device=h.device)
        k = torch.empty(R, d, dtype=torch.bfloat16, device=h.device)
        # q leaves the norm kernel already multiplied by softmax_scale * log2(e) (attention.py:96-127), in fp32
        # before its one rounding to bf16: the attention kernel's exponentials then need no per-score multiply
        q_scale = D ** -0.5 * 1.4426950408889634
        wq, wk = self._norm_w("norm_q"), self._norm_w("norm_k")          # q and k: one launch (two column segments)
        ops.rmsnorm_rope_bf16_pair_raw(ptr(qk), 2 * d, d, ptr(q), ptr(k), R, d, ptr(wq) if wq is not None else None,
                                       ptr(wk) if wk is not None else None, self.eps, int(self.qk_norm), ptr(fc.rope_cos),
                                       ptr(fc.rope_sin), fc.rope_cos.shape[0], D, ptr(fc.grid32), S, out_scale0=q_scale,
                                       out_scale1=1.0)
        del qk
        if not fused:
            # V^T[b] = Wv h_b^T + bv  ->  [B, dim, Sp]   (pad columns stay zero)
            wv, bv = self._w("v")
            ops.gemm_raw(ptr(wv), ptr(h), ptr(vt), d, S, d, d, d, Sp, EPI_BF16, bias=ptr(bv), bias_mode=BIAS_M, batch=B,
                         strideA=0, strideB=S * d, strideC=d * Sp)
        o = torch.empty(R, d, dtype=torch.bfloat16, device=h.device)
        ops.flash_attn_raw(ptr(q), ptr(k), ptr(vt), ptr(o), ptr(fc.seq_lens32), B, N, S, S, S * d, d, S * d, d,
                           d * Sp, S * d, d, Sp, D ** -0.5, q_prescaled=1)
        return o

    def forward(self, x, seq_lens, grid_sizes, freqs, _fc: Optional["_FwdCtx"] = None):
        """Reference signature (model.py:132): x [B, L, C] -> [B, L, C] fp32."""
        B, S, d = x.shape
        fc = _fc or _make_ctx_for_attention(self, x, seq_lens, grid_sizes, freqs)
        h = x.reshape(B * S, d)
        h = h if h.dtype == torch.bfloat16 else ops.cast_bf16(h.float().contiguous())
        o = self._attend(h.contiguous(), fc)
        wo, bo = self._w("o")
        return ops.gemm(o, wo, bias=bo, epilogue=EPI_F32).view(B, S, d)


class WanT2VCrossAttention(WanSelfAttention):
    """model.py:164-186."""

    def _context_kv(self, fc: "_FwdCtx", kname="k", vname="v", nname="norm_k", lo=0, hi=None):
        """K (normalised, bf16 [B*L, dim]) and V^T ([B, dim, Lp]) of context rows [lo, hi)."""
        kv = getattr(fc, "kv", None)
        if kv is not None and (id(self), kname) in kv:
            return kv[(id(self), kname)]
        B, d = fc.B, self.dim
        ctx = fc.ctx                                    # bf16 [B, Lc, dim]
        hi = fc.Lc if hi is None else hi
        L = hi - lo
        Lp = _round_up(L, 64)
        wk, bk = self._w(kname)
        wv, bv = self._w(vname)
        kf = torch.empty(B, L, d, dtype=torch.float32, device=ctx.device)
        ops.gemm_raw(ptr(ctx, lo * d), ptr(wk), ptr(kf), L, d, d, d, d, d, EPI_F32, bias=ptr(bk), bias_mode=BIAS_N,
                     batch=B, strideA=fc.Lc * d, strideB=0, strideC=L * d)
        nw = self._norm_w(nname)
        kn = ops.rmsnorm_rope(kf.view(B * L, d), nw, self.eps, do_norm=self.qk_norm)
        vt = torch.zeros(B, d, Lp, dtype=torch.bfloat16, device=ctx.device)
        ops.gemm_raw(ptr(wv), ptr(ctx, lo * d), ptr(vt), d, L, d, d, d, Lp, EPI_BF16, bias=ptr(bv), bias_mode=BIAS_M,
                     batch=B, strideA=0, strideB=fc.Lc * d, strideC=d * Lp)
        if kv is not None:
            kv[(id(self), kname)] = (kn, vt, L, Lp)
        return kn, vt, L, Lp

    def _query(self, h, fc):
        R, d = h.shape
        wq, bq = self._w("q")
        # as the self-attention's q|k: the projection leaves the GEMM in bf16 (fp32 accumulate), the norm takes its
        # statistics in fp32 from it — half the traffic of an fp32 q between the two kernels
        qb = ops.gemm(h, wq, bias=bq, epilogue=EPI_BF16)
        q = torch.empty(R, d, dtype=torch.bfloat16, device=h.device)
        w = self._norm_w("norm_q")
        ops.rmsnorm_rope_bf16_raw(ptr(qb), d, ptr(q), R, d, ptr(w) if w is not None else None, self.eps,
                                  int(self.qk_norm), None, None, 0, self.head_dim, None, fc.S)
        return q

    def _attend_ctx(self, h, fc: "_FwdCtx"):
        """Returns the list of attention outputs (bf16 [B*S, dim]) whose sum feeds the o-projection."""
        B, S, d, N, D = fc.B, fc.S, self.dim, self.num_heads, self.head_dim
        q = self._query(h, fc)
        kn, vt, L, Lp = self._context_kv(fc)
        o = torch.empty(B * S, d, dtype=torch.bfloat16, device=h.device)
        ops.flash_attn_raw(ptr(q), ptr(kn), ptr(vt), ptr(o), ptr(fc.ctx_lens32), B, N, S, L, S * d, d, L * d, d,
                           d * Lp, S * d, d, Lp, D ** -0.5)
        return [o]

    def forward(self, x, context, context_lens, _fc: Optional["_FwdCtx"] = None):
        """Reference signature (model.py:166): returns [B, L1, C] fp32."""
        B, S, d = x.shape
        fc = _fc or _make_ctx_for_cross(self, x, context, context_lens)
        h = x.reshape(B * S, d)
        h = h if h.dtype == torch.bfloat16 else ops.cast_bf16(h.float().contiguous())
        outs = self._attend_ctx(h.contiguous(), fc)
        wo, bo = self._w("o")
        y = ops.gemm(outs[0], wo, bias=bo, epilogue=EPI_F32)
        for extra in outs[1:]:
            ops.gemm(extra, wo, out=y, epilogue=ops.EPI_F32_ACCUM)
        return y.view(B, S, d)


class WanI2VCrossAttention(WanT2VCrossAttention):
    """model.py:189-230 — extra attention over the first 257 (CLIP image) context tokens."""

    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        super().__init__(dim, num_heads, window_size, qk_norm, eps)
        self.k_img = nn.Linear(dim, dim)
        self.v_img = nn.Linear(dim, dim)
        self.norm_k_img = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def _attend_ctx(self, h, fc):
        B, S, d, N, D = fc.B, fc.S, self.dim, self.num_heads, self.head_dim
        n_img = 257
        q = self._query(h, fc)
        ki, vti, Li, Lip = self._context_kv(fc, "k_img", "v_img", "norm_k_img", 0, n_img)
        kt, vtt, Lt, Ltp = self._context_kv(fc, "k", "v", "norm_k", n_img, fc.Lc)
        o_img = torch.empty(B * S, d, dtype=torch.bfloat16, device=h.device)
        ops.flash_attn_raw(ptr(q), ptr(ki), ptr(vti), ptr(o_img), None, B, N, S, Li, S * d, d, Li * d, d, d * Lip,
                           S * d, d, Lip, D ** -0.5)
        o = torch.empty(B * S, d, dtype=torch.bfloat16, device=h.device)
        # the reference passes the (text + 257) lengths here (model.py:223,537); keys are clipped to the text rows
        ops.flash_attn_raw(ptr(q), ptr(kt), ptr(vtt), ptr(o), ptr(fc.ctx_lens32), B, N, S, Lt, S * d, d, Lt * d, d,
                           d * Ltp, S * d, d, Ltp, D ** -0.5)
        return [o, o_img]


WAN_CROSSATTENTION_CLASSES = {"t2v_cross_attn": WanT2VCrossAttention, "i2v_cross_attn": WanI2VCrossAttention}


class WanAttentionBlock(nn.Module):
    """model.py:239-330."""

    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True,
                 cross_attn_norm=False, eps=1e-6):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads = dim, ffn_dim, num_heads
        self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = window_size, qk_norm, cross_attn_norm, eps
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WAN_CROSSATTENTION_CLASSES[cross_attn_type](dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self._packed = _Packed()

    def _ffn_w(self, i):
        lin = self.ffn[i]
        return self._packed.get(f"ffn{i}", (lin.weight, lin.bias),
                                lambda: (_bf16(lin.weight), lin.bias.detach().float().contiguous()))

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, block_idx=0,
                _fc: Optional["_FwdCtx"] = None, _part: str = "all"):
        """x fp32 [B, L, C] residual stream (updated and returned), e fp32 [B, 6, C].
        ``_part`` (WanModel.forward_cfg_pair): "self" = the self-attention sub-layer only, "rest" = what follows it."""
        assert e.dtype == torch.float32
        B, S, d = x.shape
        fc = _fc or _make_ctx_for_block(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens)
        R = B * S
        if not (x.dtype == torch.float32 and x.is_contiguous()):
            x = x.float().contiguous()
        mod = self.modulation.detach()
        if mod.dtype != torch.float32:
            mod = mod.float()
        e0 = fc.e0
        xp, six_d = ptr(x), 6 * d

        def ln_mod(shift_i, scale_i):
            h = torch.empty(R, d, dtype=torch.bfloat16, device=x.device)
            ops.layernorm_modulate_raw(xp, ptr(h), R, d, self.eps, 1.0, ptr(mod, scale_i * d), ptr(e0, scale_i * d),
                                       six_d, ptr(mod, shift_i * d), ptr(e0, shift_i * d), six_d, S)
            return h

        def resid(a, w, b, gate_i=None):
            # x += (a w^T + b) * gate   (gate = modulation[gate_i] + e0[:, gate_i], or 1)
            M, K = a.shape
            if gate_i is None:
                ops.gemm_raw(ptr(a), ptr(w), xp, M, d, K, K, K, d, EPI_RESID, bias=ptr(b) if b is not None else None,
                             bias_mode=BIAS_N if b is not None else BIAS_NONE, gate_const=1.0, split_k=fc.split_k)
            else:
                # (split_k: at one or two [16,1,60,104] clips the FFN-down contraction, K = ffn_dim over 56 / 104 tiles,
                # runs in slices — ABI v9; nothing is split at the sampling sizes)
                ops.gemm_raw(ptr(a), ptr(w), xp, M, d, K, K, K, d, EPI_RESID, bias=ptr(b), bias_mode=BIAS_N,
                             gate0=ptr(mod, gate_i * d), gate1=ptr(e0, gate_i * d), gate1_stride=six_d, gate_rows=S,
                             gate_const=0.0, split_k=fc.split_k)

        # ---- self-attention: x += o(attn(LN(x)(1+e1)+e0)) * e2        model.py:292-296
        if _part != "rest":
            h = ln_mod(0, 1)
            o = self.self_attn._attend(h, fc)
            wo, bo = self.self_attn._w("o")
            resid(o, wo, bo, 2)
            del h, o
            if _part == "self":
                return x
        # ---- cross-attention: x += o(attn(norm3(x), context))          model.py:313
        h = torch.empty(R, d, dtype=torch.bfloat16, device=x.device)
        if self.cross_attn_norm:
            n3 = self.norm3
            ops.layernorm_modulate_raw(xp, ptr(h), R, d, n3.eps, 0.0, ptr(n3.weight.detach().float()), None, 0,
                                       ptr(n3.bias.detach().float()), None, 0, R)
        else:
            ops.cast_bf16(x.view(R, d), out=h)
        outs = self.cross_attn._attend_ctx(h, fc)
        wo, bo = self.cross_attn._w("o")
        resid(outs[0], wo, bo)
        for extra in outs[1:]:
            resid(extra, wo, None)
        del h, outs
        # ---- FFN: x += (W2 gelu(W1 (LN(x)(1+e4)+e3) + b1) + b2) * e5    model.py:314-328
        h = ln_mod(3, 4)
        w1, b1 = self._ffn_w(0)
        w2, b2 = self._ffn_w(2)
        u = torch.empty(R, self.ffn_dim, dtype=torch.bfloat16, device=x.device)
        ops.gemm_raw(ptr(h), ptr(w1), ptr(u), R, self.ffn_dim, d, d, d, self.ffn_dim, EPI_GELU_BF16, bias=ptr(b1),
                     bias_mode=BIAS_N)
        resid(u, w2, b2, 5)
        return x

    def cross_attn_ffn(self, x, context, context_lens, e, block_idx):  # pragma: no cover - API parity
        raise NotImplementedError("cross_attn_ffn is fused into WanAttentionBlock.forward in this build")


class Head(nn.Module):
    """model.py:332-359."""

    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        out_dim = math.prod(patch_size) * out_dim
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)
        self._packed = _Packed()

    def forward(self, x, e):
        """x fp32 [B, L, C], e fp32 [B, C] -> fp32 [B, L, prod(patch)*out_dim]."""
        assert e.dtype == torch.float32
        B, S, d = x.shape
        R = B * S
        x = x.float().contiguous()
        e = e.contiguous()
        mod = self.modulation.detach().float()
        h = torch.empty(R, d, dtype=torch.bfloat16, device=x.device)
        ops.layernorm_modulate_raw(ptr(x), ptr(h), R, d, self.eps, 1.0, ptr(mod, d), ptr(e), d, ptr(mod, 0), ptr(e), d,
                                   S)
        w, b = self._packed.get("head", (self.head.weight, self.head.bias), lambda: (
            _bf16(self.head.weight), self.head.bias.detach().float().contiguous()))
        return ops.gemm(h, w, bias=b, epilogue=EPI_F32).view(B, S, -1)


class MLPProj(nn.Module):
    """model.py:362-374 (CLIP image tokens -> model dim, i2v only)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, in_dim), nn.GELU(),
                                  nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim))
        self._packed = _Packed()

    def _w(self, i):
        lin = self.proj[i]
        return self._packed.get(f"l{i}", (lin.weight, lin.bias),
                                lambda: (_bf16(lin.weight), lin.bias.detach().float().contiguous()))

    def forward(self, image_embeds):
        """[B, 257, 1280] CLIP tokens -> bf16 [B, 257, out_dim]  (LayerNorm, Linear, GELU(erf), Linear, LayerNorm)."""
        B, L, Cin = image_embeds.shape
        x = image_embeds.float().contiguous().view(B * L, Cin)
        ln0, ln4 = self.proj[0], self.proj[4]
        h = ops.layernorm_modulate(x, ln0.eps, 0.0, mul0=ln0.weight.detach().float(), add0=ln0.bias.detach().float())
        w1, b1 = self._w(1)
        w3, b3 = self._w(3)
        h = ops.gemm(h, w1, bias=b1, epilogue=ops.EPI_GELU_ERF_BF16)
        h = ops.gemm(h, w3, bias=b3, epilogue=EPI_F32)
        out = ops.layernorm_modulate(h, ln4.eps, 0.0, mul0=ln4.weight.detach().float(),
                                     add0=ln4.bias.detach().float())
        return out.view(B, L, -1)


def _rope_tables(freqs: torch.Tensor, device):
    """fp32 cos/sin of the reference's complex rotary table."""
    return (freqs.real.to(torch.float32).contiguous().to(device),
            freqs.imag.to(torch.float32).contiguous().to(device))


def _make_ctx_for_block(block, x, e, seq_lens, grid_sizes, freqs, context, context_lens):
    """Build the per-forward context when a block is called stand-alone with
    the reference's positional arguments (e.g. from a test or a hook)."""
    B, S, d = x.shape
    fc = _FwdCtx()
    fc.B, fc.S, fc.dim = B, S, d
    fc.split_k = True
    fc.e0 = e.contiguous()
    fc.seq_lens32 = seq_lens.to(device=x.device, dtype=torch.int32).contiguous()
    fc.grid32 = grid_sizes.to(device=x.device, dtype=torch.int32).contiguous()
    fc.rope_cos, fc.rope_sin = freqs if isinstance(freqs, tuple) else _rope_tables(freqs, x.device)
    ctx = context
    fc.ctx = ctx if ctx.dtype == torch.bfloat16 else ops.cast_bf16(ctx.float().contiguous())
    fc.Lc = ctx.shape[1]
    fc.ctx_lens32 = (context_lens.to(device=x.device, dtype=torch.int32).contiguous()
                     if context_lens is not None else None)
    return fc


def _make_ctx_for_attention(attn, x, seq_lens, grid_sizes, freqs):
    B, S, d = x.shape
    fc = _FwdCtx()
    fc.B, fc.S, fc.dim = B, S, d
    fc.seq_lens32 = seq_lens.to(device=x.device, dtype=torch.int32).contiguous()
    fc.grid32 = grid_sizes.to(device=x.device, dtype=torch.int32).contiguous()
    fc.rope_cos, fc.rope_sin = freqs if isinstance(freqs, tuple) else _rope_tables(freqs, x.device)
    return fc


def _make_ctx_for_cross(attn, x, context, context_lens):
    B, S, d = x.shape
    fc = _FwdCtx()
    fc.B, fc.S, fc.dim = B, S, d
    fc.ctx = context if context.dtype == torch.bfloat16 else ops.cast_bf16(context.float().contiguous())
    fc.Lc = context.shape[1]
    fc.ctx_lens32 = (context_lens.to(device=x.device, dtype=torch.int32).contiguous()
                     if context_lens is not None else None)
    return fc


class WanModel(nn.Module):
    r"""Wan diffusion backbone (t2v / i2v) — reference model.py:377-612."""

    ignore_for_config = ["patch_size", "cross_attn_norm", "qk_norm", "text_dim", "window_size"]
    _no_split_modules = ["WanAttentionBlock"]

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, use_checkpoint=True):
        super().__init__()
        assert model_type in ["t2v", "i2v"]
        self.config = dict(model_type=model_type, patch_size=tuple(patch_size), text_len=text_len, in_dim=in_dim,
                           dim=dim, ffn_dim=ffn_dim, freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim,
                           num_heads=num_heads, num_layers=num_layers, window_size=tuple(window_size),
                           qk_norm=qk_norm, cross_attn_norm=cross_attn_norm, eps=eps)
        self.model_type = model_type
        self.use_checkpoint = use_checkpoint
        # how a True flag is honoured (model_train.py): "auto" keeps the activations when a step's worth of them fits in
        # half of the free HBM (same gradients, no second forward pass), "always" re-runs every block in the backward
        self.checkpoint_policy = "auto"
        # reference quirk (model.py:317-324): FFNs of blocks > 10 receive no gradient; False = full gradients
        self.reference_ffn_freeze = True
        # False (default): products with few rows and a long contraction — the FFN-down projection at one or two
        # [16,1,60,104] clips — are summed in 2-4 k slices (ABI v9), so a sample's last bits depend on how many samples
        # share its forward (a clip alone, in a CFG pair, in a batch of four: 4 / 2 / no slices; 9e-3 between the
        # guided velocities, where each is 1e-2 from the fp32 arithmetic).  True: every product keeps one summation order
        # whatever the batch — bit-identical outputs for a sample alone and inside any batch, at the price of the
        # small-M speed-up (15.3 -> 16.1 ms per single-frame CFG pair).  Nothing is split at the sampling sizes
        # (S = 32 760) either way.  ADVICE round 4; pinned by test_batch_invariant_flag_pins_the_summation_order.
        self.batch_invariant = False
        self.patch_size = tuple(patch_size)
        self.text_len, self.in_dim, self.dim, self.ffn_dim = text_len, in_dim, dim, ffn_dim
        self.freq_dim, self.text_dim, self.out_dim = freq_dim, text_dim, out_dim
        self.num_heads, self.num_layers, self.window_size = num_heads, num_layers, window_size
        self.qk_norm, self.cross_attn_norm, self.eps = qk_norm, cross_attn_norm, eps

        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"),
                                            nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))

        cross_attn_type = "t2v_cross_attn" if model_type == "t2v" else "i2v_cross_attn"
        self.blocks = nn.ModuleList([
            WanAttentionBlock(cross_attn_type, dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps)
            for _ in range(num_layers)])
        self.head = Head(dim, out_dim, self.patch_size, eps)

        assert (dim % num_heads) == 0 and (dim // num_heads) % 2 == 0
        d = dim // num_heads
        if d != 128:
            raise NotImplementedError(f"head_dim {d}: the gfx950 attention kernel is built for head_dim 128 "
                                      "(every Wan2.1 checkpoint)")
        # not a buffer, as in the reference (model.py:484-492)
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        if model_type == "i2v":
            self.img_emb = MLPProj(1280, dim)
        self._packed = _Packed()
        self._rope_dev = None
        self.init_weights()

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_config(cls, config: dict, **kw):
        cfg = {k: v for k, v in config.items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, checkpoint_dir: str, **kw):
        """diffusers-layout loader: ``config.json`` + ``diffusion_pytorch_model*.safetensors``
        (what the reference gets from diffusers' ModelMixin, text2video.py:86)."""
        import glob
        import json
        import os
        from safetensors.torch import load_file
        with open(os.path.join(checkpoint_dir, "config.json")) as fh:
            config = json.load(fh)
        allowed = set(cls.__init__.__code__.co_varnames)
        model = cls(**{k: v for k, v in config.items() if k in allowed}, **kw)
        sd = {}
        for f in sorted(glob.glob(os.path.join(checkpoint_dir, "diffusion_pytorch_model*.safetensors"))):
            sd.update(load_file(f))
        model.load_state_dict(sd, strict=True)
        return model

    # ------------------------------------------------------------------ forward
    def _rope(self, device):
        if self._rope_dev is None or self._rope_dev[0] != str(device):
            self._rope_dev = (str(device),) + _rope_tables(self.freqs, device)
        return self._rope_dev[1], self._rope_dev[2]

    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, extra_conditions=None):
        r"""Same contract as the reference (model.py:502-563):
        x: list (or batched tensor) of [C_in, F, H, W]; t: [B]; context: list of [L, text_dim];
        returns list of fp32 [C_out, F, H, W].

        ``extra_conditions`` is the keyword Omnihuman/omnihuman_wan_t2v.py:408-414 passes (the reference model does
        not accept it): a ``[B, Ne, dim]`` tensor of condition tokens in model width, or a dict holding it under
        ``"tokens"``.  They are prepended to the embedded text context (t2v, inference), so every block's
        cross-attention attends to them — see omnihuman_wan_t2v.OmniHumanWanT2V.condition_tokens."""
        device = self.patch_embedding.weight.device
        if device.type != "cuda":
            raise ops.OmhError("WanModel.forward runs on the MI355X only (no CPU fallback): move the model "
                               "to a GPU device")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if isinstance(context, ContextState):
                raise ValueError("a ContextState is an inference-time cache: pass the raw context when training")
            from .model_train import forward_train
            return forward_train(self, x, t, context, seq_len, clip_fea, y, extra_conditions)
        with torch.no_grad():
            return self._forward_infer(x, t, context, seq_len, clip_fea, y, extra_conditions)

    def _embed(self, x, t, context, seq_len, clip_fea, y, extra_conditions=None):
        device = self.patch_embedding.weight.device
        d = self.dim
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        x = [u.to(device=device, dtype=torch.float32).contiguous() for u in x]
        B = len(x)
        pt, ph, pw = self.patch_size
        grids = [(u.shape[1] // pt, u.shape[2] // ph, u.shape[3] // pw) for u in x]
        lens = [g[0] * g[1] * g[2] for g in grids]
        assert max(lens) <= seq_len, f"Max seq len {max(lens)} exceeds limit {seq_len}"
        # ---- patch embedding (Conv3d k=s=patch as a GEMM); padded rows stay zero  model.py:515-522
        kin = self.in_dim * pt * ph * pw
        Kp = _round_up(kin, 8)
        wpe, bpe = self._packed.get("patch", (self.patch_embedding.weight, self.patch_embedding.bias), lambda: (
            _bf16(torch.nn.functional.pad(self.patch_embedding.weight.detach().flatten(1).float(), (0, Kp - kin))),
            self.patch_embedding.bias.detach().float().contiguous()))
        xs = torch.empty(B, seq_len, d, dtype=torch.float32, device=device)
        for b, u in enumerate(x):
            assert u.shape[0] == self.in_dim, f"expected {self.in_dim} input channels, got {u.shape[0]}"
            tok = ops.patchify(u, self.patch_size, Kp)
            ops.gemm_raw(ptr(tok), ptr(wpe), ptr(xs, b * seq_len * d), lens[b], d, Kp, Kp, Kp, d, EPI_F32,
                         bias=ptr(bpe), bias_mode=BIAS_N)
            if lens[b] < seq_len:
                xs[b, lens[b]:].zero_()                 # only the padding rows need the fill; the GEMM wrote the rest
        # ---- time embedding (fp32)                                              model.py:526-528
        t = t.to(device=device)
        te0, te2, tp1 = self.time_embedding[0], self.time_embedding[2], self.time_projection[1]
        sin = ops.sinusoidal_embedding(t, self.freq_dim)
        e = ops.dense_f32(ops.dense_f32(sin, te0.weight.detach().float(), te0.bias.detach().float(), 0, 1),
                          te2.weight.detach().float(), te2.bias.detach().float(), 0, 0)
        e0 = ops.dense_f32(e, tp1.weight.detach().float(), tp1.bias.detach().float(), 1, 0).view(B, 6, d)
        fc = _FwdCtx()
        fc.B, fc.S, fc.dim = B, seq_len, d
        fc.split_k = not getattr(self, "batch_invariant", False)
        fc.e0 = e0.contiguous()
        fc.seq_lens32 = _dev_ints(lens, torch.int32, device)
        fc.grid32 = _dev_ints(grids, torch.int32, device)
        fc.rope_cos, fc.rope_sin = self._rope(device)
        fc.seq_lens_host = list(lens)                                       # host copies: no device sync later
        ctx_lens = self._attach_context(fc, context, clip_fea, extra_conditions)
        return xs, e, fc, grids, lens, ctx_lens

    def _attach_context(self, fc, context, clip_fea=None, extra_conditions=None):
        """The context-dependent fields of a forward's shared state: text (+ image) embedding, model.py:531-537."""
        device = self.patch_embedding.weight.device
        state = context if isinstance(context, ContextState) else None
        if state is not None:
            if state.model_id != id(self) or state.version != self._context_signature() or state.B != fc.B:
                raise ValueError("ContextState does not belong to this model / batch or the weights changed since "
                                 "encode_context(): call encode_context() again")
            ctx, ctx_lens = state.ctx, list(state.ctx_lens)
            if extra_conditions is not None:
                raise ValueError("pass extra condition tokens to encode_context(), not next to a ContextState")
        else:
            ctx, ctx_lens = self._embed_context(context, clip_fea, extra_conditions)
        fc.ctx_lens32 = _dev_ints(ctx_lens, torch.int32, device)
        fc.ctx, fc.Lc = ctx, ctx.shape[1]
        fc.kv = state.kv if state is not None else None
        fc.ctx_lens_host = list(ctx_lens)
        return ctx_lens

    def _embed_context(self, context, clip_fea=None, extra_tokens=None):
        """text_embedding (and img_emb) of zero-padded contexts: bf16 [B, (257+ | Ne+)text_len, dim] and true lengths."""
        device = self.patch_embedding.weight.device
        d, B = self.dim, len(context)
        ctx_lens = [int(u.shape[0]) for u in context]
        ctx_in = torch.zeros(B, self.text_len, self.text_dim, dtype=torch.float32, device=device)
        for b, u in enumerate(context):
            ctx_in[b, :u.shape[0]] = u.to(device=device, dtype=torch.float32)
        w0, b0 = self._packed.get("text0", (self.text_embedding[0].weight, self.text_embedding[0].bias), lambda: (
            _bf16(self.text_embedding[0].weight), self.text_embedding[0].bias.detach().float().contiguous()))
        w2, b2 = self._packed.get("text2", (self.text_embedding[2].weight, self.text_embedding[2].bias), lambda: (
            _bf16(self.text_embedding[2].weight), self.text_embedding[2].bias.detach().float().contiguous()))
        cin = ops.cast_bf16(ctx_in).view(B * self.text_len, self.text_dim)
        ctx = ops.gemm(ops.gemm(cin, w0, bias=b0, epilogue=EPI_GELU_BF16), w2, bias=b2, epilogue=EPI_BF16)
        ctx = ctx.view(B, self.text_len, d)
        if clip_fea is not None:
            ctx_img = self.img_emb(clip_fea.to(device))                    # bf16 [B, 257, dim]
            ctx = torch.cat([ctx_img, ctx], dim=1).contiguous()
            ctx_lens = [c + ctx_img.shape[1] for c in ctx_lens]
        if extra_tokens is not None:
            tok = extra_tokens.get("tokens") if isinstance(extra_tokens, dict) else extra_tokens
            if tok is not None:
                if self.model_type != "t2v":
                    raise NotImplementedError("extra condition tokens are defined for the t2v backbone")
                if tok.dim() != 3 or tok.shape[0] != B or tok.shape[2] != d:
                    raise ValueError(f"extra condition tokens must be [B={B}, Ne, dim={d}], got {tuple(tok.shape)}")
                tokb = ops.cast_bf16(tok.to(device=device, dtype=torch.float32).contiguous())
                ctx = torch.cat([tokb, ctx], dim=1).contiguous()
                ctx_lens = [c + tok.shape[1] for c in ctx_lens]
        return ctx, ctx_lens

    def _context_signature(self):
        """Identity + version of every parameter a ContextState depends on."""
        mods = [self.text_embedding]
        if hasattr(self, "img_emb"):
            mods.append(self.img_emb)
        ps = [p for m in mods for p in m.parameters()]
        for blk in self.blocks:
            ca = blk.cross_attn
            for name in ("k", "v", "norm_k", "k_img", "v_img", "norm_k_img"):
                m = getattr(ca, name, None)
                if isinstance(m, nn.Module):
                    ps.extend(m.parameters())
        return tuple((p.data_ptr(), p._version) for p in ps)

    @torch.no_grad()
    def encode_context(self, context, clip_fea=None, extra_conditions=None) -> "ContextState":
        """Pre-compute what depends only on (context, clip_fea, extra_conditions).  Pass the result as ``context`` to
        forward() (inference only); every block adds its cross-attention K / V^T on first use and reuses them
        afterwards."""
        st = ContextState()
        st.ctx, st.ctx_lens = self._embed_context(context, clip_fea, extra_conditions)
        st.kv, st.B, st.model_id, st.version = {}, len(context), id(self), self._context_signature()
        return st

    def _forward_infer(self, x, t, context, seq_len, clip_fea=None, y=None, extra_conditions=None):
        xs, e, fc, grids, lens, ctx_lens = self._embed(x, t, context, seq_len, clip_fea, y, extra_conditions)
        seq_lens = _dev_ints(lens, torch.long, xs.device)
        grid_sizes = _dev_ints(grids, torch.long, xs.device)
        context_lens = _dev_ints(ctx_lens, torch.long, xs.device)
        freqs = (fc.rope_cos, fc.rope_sin)
        for i, block in enumerate(self.blocks):
            # module __call__ so forward hooks on blocks fire (seaweed_apt/model.py:150-155)
            xs = block(xs, fc.e0, seq_lens, grid_sizes, freqs, fc.ctx, context_lens, block_idx=i, _fc=fc)
        out = self.head(xs, e)                                   # fp32 [B, seq_len, prod(patch)*out_dim]
        return self.unpatchify(out, grid_sizes, _grids=grids)

    def forward_cfg_pair(self, x, t, context, context_null, seq_len, clip_fea=None, y=None):
        """The two forwards of a classifier-free-guided step — ``forward(x, t, context, ...)`` and ``forward(x, t,
        context_null, ...)`` on the SAME latents and timestep (text2video.py:238-241, generate.py:205-229) — with what
        they share computed once: the patch / time embedding and block 0's self-attention sub-layer (LayerNorm + modulate,
        q | k | v, RoPE, self-attention, o-projection: its inputs are x and t alone; the context enters at block 0's
        cross-attention).  Everything after that point runs per branch, with the launches of a batch-of-one forward, so
        both results equal the two separate calls BIT FOR BIT (the kernels are deterministic); at S = 32 760 it is one
        self-attention launch of 60 less per step.  Inference only (a model whose parameters require grad must be called
        under torch.no_grad(): the training step has no such pair); contexts may be ContextStates.
        Returns (list of cond outputs, list of uncond outputs)."""
        device = self.patch_embedding.weight.device
        if device.type != "cuda":
            raise ops.OmhError("WanModel.forward_cfg_pair runs on the MI355X only (no CPU fallback): move the model "
                               "to a GPU device")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("forward_cfg_pair is an inference path: call it under torch.no_grad() or on a model "
                               "with requires_grad_(False) (training goes through forward())")
        with torch.no_grad():
            return self._forward_cfg_pair(x, t, context, context_null, seq_len, clip_fea, y)

    def _forward_cfg_pair(self, x, t, context, context_null, seq_len, clip_fea, y):
        xs, e, fc, grids, lens, ctx_lens = self._embed(x, t, context, seq_len, clip_fea, y)
        seq_lens = _dev_ints(lens, torch.long, xs.device)
        grid_sizes = _dev_ints(grids, torch.long, xs.device)
        freqs = (fc.rope_cos, fc.rope_sin)
        blk0 = self.blocks[0]
        # (no module __call__ here: a forward hook on block 0 fires once per branch, below, with the block's full output)
        xs = blk0.forward(xs, fc.e0, seq_lens, grid_sizes, freqs, fc.ctx, None, block_idx=0, _fc=fc, _part="self")
        fc_u = _FwdCtx()
        for name in _FwdCtx.__slots__:
            if hasattr(fc, name):
                setattr(fc_u, name, getattr(fc, name))
        self._attach_context(fc_u, context_null, clip_fea)
        outs = []
        for f, xb in ((fc, xs.clone()), (fc_u, xs)):
            context_lens = _dev_ints(f.ctx_lens_host, torch.long, xs.device)
            for i, block in enumerate(self.blocks):
                xb = block(xb, f.e0, seq_lens, grid_sizes, freqs, f.ctx, context_lens, block_idx=i, _fc=f,
                           _part="rest" if i == 0 else "all")
            outs.append(self.unpatchify(self.head(xb, e), grid_sizes, _grids=grids))
        return outs[0], outs[1]

    def unpatchify(self, x, grid_sizes, _grids=None):
        """model.py:565-588 — 'fhwpqrc->cfphqwr' per sample."""
        grids = _grids if _grids is not None else [tuple(int(a) for a in g) for g in grid_sizes.tolist()]
        return [ops.unpatchify(u.contiguous(), self.out_dim, g, self.patch_size) for u, g in zip(x, grids)]

    def init_weights(self):
        """model.py:590-612."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for m in self.text_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        for m in self.time_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        nn.init.zeros_(self.head.head.weight)
