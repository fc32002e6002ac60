"""umT5 text encoder of the Wan pipelines on the gfx950 kernels — the call surface of
``seaweed_apt/wan/modules/t5.py`` (``T5Encoder``, ``umt5_xxl``, ``T5EncoderModel``) with the same module tree and
state-dict keys, so ``models_t5_umt5-xxl-enc-bf16.pth`` loads (text2video.py:64-70).

Arithmetic (include/omh.h, "Prompt-side encoders"): embedding lookup ``omh_gather_rows_f32``, T5LayerNorm
``omh_rmsnorm_f32``, every Linear on ``omh_gemm_bf16`` (bf16 operands, fp32 accumulate, fp32 residual stream),
attention over the 64 heads of width 64 as two head-batched GEMMs around ``omh_softmax_bias_rows`` (relative-position
bias + key mask + softmax in one pass; T5 does not scale the scores, t5.py:112), gated GELU on the GEMM epilogue +
``omh_mul_bf16``.  The encoder runs once per prompt; nothing here is on the per-step path.

``reference_block_quirk`` (default True) keeps this repository's cut-down ``T5SelfAttention.forward``
(t5.py:166-176): ``x = norm1(x); x = x + attn(x)`` — residual on the NORMALISED stream, no feed-forward.  False runs
the upstream umT5 block the checkpoint was trained with.
"""
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .._backend import ops
from .model import _Packed, _bf16, _round_up

__all__ = ["T5Encoder", "T5EncoderModel", "umt5_xxl", "relative_position_buckets"]

EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID = ops.EPI_BF16, ops.EPI_F32, ops.EPI_GELU_BF16, ops.EPI_RESID
ptr = ops.ptr


def relative_position_buckets(lq: int, lk: int, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (t5.py:244-268): int32 [lq, lk] on the host
    (integer / log arithmetic on L^2 indices that depends on the length alone — host logic, like the samplers'
    sigma schedules)."""
    rel = torch.arange(lk)[None, :] - torch.arange(lq)[:, None]
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (out + torch.where(rel < max_exact, rel, large)).to(torch.int32)


def encoder_attention(q, k, vt, H: int, Dh: int, L: int, scale: float, bucket=None, table=None, klen=None):
    """softmax(q k^T * scale + bias) v for ONE sample and all heads: q, k bf16 [L, H*Dh]; vt bf16 [H*Dh, Lp] (V
    transposed, pad columns zero).  Two head-batched GEMMs around the fused bias / mask / softmax kernel."""
    Lp = vt.shape[1]
    d = H * Dh
    s = torch.empty(H * L, Lp, dtype=torch.float32, device=q.device)
    ops.gemm_raw(ptr(q), ptr(k), ptr(s), L, L, Dh, d, d, Lp, EPI_F32, batch=H, strideA=Dh, strideB=Dh, strideC=L * Lp)
    p = ops.softmax_bias_rows(s, H, L, scale, bucket, table, klen, ldy=Lp)
    o = torch.empty(L, d, dtype=torch.bfloat16, device=q.device)
    ops.gemm_raw(ptr(p), ptr(vt), ptr(o), L, Dh, Lp, Lp, Lp, d, EPI_BF16, batch=H, strideA=L * Lp, strideB=Dh * Lp,
                 strideC=Dh)
    return o


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads, dropout=0.1):
        assert dim_attn % num_heads == 0
        super().__init__()
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn, dropout=0.1):
        super().__init__()
        self.dim, self.dim_ffn = dim, dim_ffn
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False), nn.Identity())   # [1] = GELU-tanh (no parameters)
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets, num_heads, bidirectional, max_dist=128):
        super().__init__()
        self.num_buckets, self.num_heads, self.bidirectional, self.max_dist = num_buckets, num_heads, bidirectional, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.shared_pos = shared_pos
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads, dropout)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn, dropout)
        self.pos_embedding = None if shared_pos else T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True)


class T5Encoder(nn.Module):
    """t5.py:272-322 (constructor and state-dict keys as there)."""

    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True) if shared_pos else None
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos, dropout)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)
        self.reference_block_quirk = True
        self._packed = _Packed()
        self._buckets = {}

    def _w(self, lin: nn.Linear, key):
        return self._packed.get(key, (lin.weight,), lambda: _bf16(lin.weight))

    def _bucket_table(self, L, device):
        b = self._buckets.get((L, str(device)))
        if b is None:
            b = self._buckets[(L, str(device))] = relative_position_buckets(L, L, self.num_buckets).to(device)
        return b

    @torch.no_grad()
    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        """Token embedding (t5.py:306): ids int64 [B, L] -> fp32 [B, L, dim].  A bf16 table is read as it is."""
        dev = self.token_embedding.weight.device
        emb = self.token_embedding.weight.detach()
        if emb.dtype not in (torch.float32, torch.bfloat16):
            emb = emb.float()
        return ops.gather_rows(emb.contiguous(), ids.to(dev))

    @torch.no_grad()
    def run_layers(self, xb: torch.Tensor, klen: int, lo: int = 0, hi: Optional[int] = None) -> torch.Tensor:
        """Blocks [lo, hi) on ONE sample's fp32 residual stream [L, dim] with ``klen`` unmasked (leading) keys
        (T5SelfAttention.forward, t5.py:166-176 / the upstream block).  The stream stays fp32 whatever the parameter
        dtype: the reference casts it to the weights' bf16 in every T5LayerNorm (t5.py:66-68) — this path is the more
        precise of the two; parameters are used in bf16 (the MFMA operand type) either way."""
        dev = xb.device
        L, d = xb.shape
        H, Dh = self.num_heads, self.dim_attn // self.num_heads
        Lp = _round_up(L, 8)
        bucket = self._bucket_table(L, dev)
        f32 = lambda w: w.detach() if w.dtype == torch.float32 else w.detach().float()
        xb = xb.contiguous().clone()
        for li in range(lo, len(self.blocks) if hi is None else hi):
            blk = self.blocks[li]
            table = f32((self.pos_embedding if self.shared_pos else blk.pos_embedding).embedding.weight).contiguous()
            nf, nb_ = ops.rmsnorm_f32(xb, f32(blk.norm1.weight), blk.norm1.eps, want_f32=self.reference_block_quirk,
                                      want_bf16=True)
            a = blk.attn
            wq, wk, wv, wo = (self._w(getattr(a, n), (li, n)) for n in ("q", "k", "v", "o"))
            q = ops.gemm(nb_, wq)
            k = ops.gemm(nb_, wk)
            vt = torch.zeros(self.dim_attn, Lp, dtype=torch.bfloat16, device=dev) if Lp != L else \
                torch.empty(self.dim_attn, Lp, dtype=torch.bfloat16, device=dev)
            ops.gemm_raw(ptr(wv), ptr(nb_), ptr(vt), self.dim_attn, L, d, d, d, Lp, EPI_BF16)     # V^T = Wv n^T
            o = encoder_attention(q, k, vt, H, Dh, L, 1.0, bucket, table, klen)
            if self.reference_block_quirk:                              # t5.py:166-176: x = norm1(x) + attn(norm1(x))
                xb = nf
            ops.gemm_raw(ptr(o), ptr(wo), ptr(xb), L, d, self.dim_attn, self.dim_attn, self.dim_attn, d, EPI_RESID,
                         gate_const=1.0)
            if not self.reference_block_quirk:                          # upstream: x += fc2(fc1(h) * gelu(gate(h)))
                _, h = ops.rmsnorm_f32(xb, f32(blk.norm2.weight), blk.norm2.eps, want_f32=False)
                f = blk.ffn
                g = ops.gemm(h, self._w(f.gate[0], (li, "gate")), epilogue=EPI_GELU_BF16)
                u = ops.gemm(h, self._w(f.fc1, (li, "fc1")))
                gu = ops.mul_bf16(u, g)
                ops.gemm_raw(ptr(gu), ptr(self._w(f.fc2, (li, "fc2"))), ptr(xb), L, d, self.dim_ffn, self.dim_ffn,
                             self.dim_ffn, d, EPI_RESID, gate_const=1.0)
        return xb

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids int64 [B, L], mask [B, L] (1 = token) -> fp32 [B, L, dim] (t5.py:305-322, eval mode)."""
        dev = self.token_embedding.weight.device
        ids = ids.to(dev)
        B, L = ids.shape
        x = self.embed(ids)                                                 # fp32 [B, L, dim]
        klens = [L] * B if mask is None else [int(v) for v in mask.to(dev).gt(0).sum(dim=1).tolist()]
        if mask is not None:                                                # the kernel masks keys >= klen: prefix masks only
            m = mask.to(dev).gt(0)
            assert bool((m == (torch.arange(L, device=dev)[None, :] < m.sum(1, keepdim=True))).all()), \
                "attention masks must be prefixes (tokens first, padding after): tokenizers pad on the right"
        out = torch.empty(B, L, self.dim, dtype=torch.float32, device=dev)
        nw = self.norm.weight.detach()
        nw = nw if nw.dtype == torch.float32 else nw.float()
        for b in range(B):
            xb = self.run_layers(x[b], klens[b])
            yf, _ = ops.rmsnorm_f32(xb, nw, self.norm.eps, want_bf16=False)
            out[b] = yf
        return out


def umt5_xxl(encoder_only=True, dtype=torch.float32, device="cpu", **kwargs):
    """t5.py:466-479 (encoder only: the decoder is never used by the pipelines)."""
    assert encoder_only, "only the encoder of umT5-XXL is part of the Wan pipelines"
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32,
               shared_pos=False, dropout=0.1)
    cfg.update(kwargs)
    with torch.device(device):
        m = T5Encoder(**cfg)
    return m.to(dtype=dtype, device=device)


class T5EncoderModel:
    """t5.py:481-528: ``T5EncoderModel(text_len, dtype, device, checkpoint_path, tokenizer_path)``;
    ``__call__(texts, device) -> list of [L_i, 4096]`` (padding stripped).  Tokenisation is the HuggingFace
    tokenizer at ``tokenizer_path`` (host side, as in the reference); ``tokenizer=`` injects any callable
    ``(list[str]) -> (ids [B, text_len] int64, mask [B, text_len])`` instead."""

    def __init__(self, text_len, dtype=torch.bfloat16, device=None, checkpoint_path=None, tokenizer_path=None,
                 shard_fn=None, tokenizer=None, model: Optional[T5Encoder] = None):
        self.text_len, self.dtype, self.checkpoint_path, self.tokenizer_path = text_len, dtype, checkpoint_path, tokenizer_path
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if shard_fn is not None:
            raise NotImplementedError("FSDP sharding of the text encoder: 9 GB of bf16 weights fit one MI355X")
        if model is None:
            # the parameters take the dtype the caller names, as in the reference (t5.py:493-500: bf16 by default,
            # 11 GB for umT5-XXL): a bf16 weight IS its MFMA operand copy (model._bf16 returns it as it is)
            model = umt5_xxl(encoder_only=True, dtype=dtype, device=self.device)
            if checkpoint_path is not None:
                model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"))
        self.model = model.eval().requires_grad_(False).to(self.device)
        self.tokenizer = tokenizer
        if tokenizer is None and tokenizer_path is not None:
            self.tokenizer = _hf_tokenizer(tokenizer_path, text_len)

    def __call__(self, texts: List[str], device=None):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: pass tokenizer_path= (HuggingFace umt5-xxl files) or tokenizer=callable")
        ids, mask = self.tokenizer(texts)
        ids, mask = ids.to(self.device), mask.to(self.device)
        seq_lens = mask.gt(0).sum(dim=1).long()
        context = self.model(ids, mask)
        dev = self.device if device is None else device
        return [u[:v].to(dev) for u, v in zip(context, seq_lens)]


def _hf_tokenizer(path, seq_len):
    """The reference's HuggingfaceTokenizer(name, seq_len, clean='whitespace') (tokenizers.py:38-82) as a closure."""
    import html
    import re
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path)

    def clean(t):
        try:
            import ftfy
            t = ftfy.fix_text(t)
        except ImportError:
            pass
        t = html.unescape(html.unescape(t)).strip()
        return re.sub(r"\s+", " ", t).strip()

    def call(texts):
        if isinstance(texts, str):
            texts = [texts]
        enc = tok([clean(t) for t in texts], return_tensors="pt", padding="max_length", truncation=True,
                  max_length=seq_len, add_special_tokens=True)
        return enc.input_ids, enc.attention_mask
    return call
