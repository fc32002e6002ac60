"""Vision tower of the XLM-Roberta CLIP (ViT-H/14) on the gfx950 kernels — what ``WanI2V`` uses of
``seaweed_apt/wan/modules/clip.py``: ``CLIPModel(dtype, device, checkpoint_path, tokenizer_path).visual(videos)``
(clip.py:501-542) = bicubic resize to 224x224, CLIP normalisation, ``VisionTransformer.forward(use_31_block=True)``
(clip.py:275-301): patch embedding, class token + position embedding, pre-LayerNorm, the first 31 of the 32
pre-norm blocks -> ``[B, 257, 1280]`` tokens for ``WanModel``'s ``img_emb`` (model.py:534-537).

``VisionTransformer`` keeps the reference's parameter names (``cls_embedding, pos_embedding, patch_embedding,
pre_norm, transformer.{i}.{norm1,attn.to_qkv,attn.proj,norm2,mlp.0,mlp.2}, post_norm, head``), so the ``visual.*``
entries of ``models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth`` load; the text tower of that checkpoint is
never evaluated by the pipelines and is skipped.

Arithmetic: ``omh_patchify`` + ``omh_gemm_bf16`` (patch embedding), ``omh_vit_embed``, ``omh_layernorm_f32`` /
``omh_layernorm_modulate`` (affine LayerNorms), ``omh_gemm_bf16`` with bias / erf-GELU / residual epilogues,
attention over 16 heads of width 80 via ``t5.encoder_attention``.  The bicubic resize stays on torch (host-side
preprocessing, as in the reference's transforms).
"""
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._backend import ops
from .model import _Packed, _bf16, _round_up
from .t5 import encoder_attention

__all__ = ["VisionTransformer", "CLIPModel", "clip_xlm_roberta_vit_h_14"]

EPI_BF16, EPI_F32, EPI_RESID, EPI_GELU_ERF = ops.EPI_BF16, ops.EPI_F32, ops.EPI_RESID, ops.EPI_GELU_ERF_BF16
ptr = ops.ptr
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _SelfAttention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.to_qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)


class _AttentionBlock(nn.Module):
    def __init__(self, dim, mlp_ratio, num_heads, norm_eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn = _SelfAttention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=norm_eps)
        self.mlp = nn.Sequential(nn.Linear(dim, int(dim * mlp_ratio)), nn.Identity(), nn.Linear(int(dim * mlp_ratio), dim),
                                 nn.Identity())                       # [1] = GELU (erf), [3] = Dropout: no parameters


class VisionTransformer(nn.Module):
    """clip.py:209-301 with pool_type='token', pre_norm=True, post_norm=False, activation='gelu' (the ViT-H/14 of
    clip_xlm_roberta_vit_h_14, clip.py:468-495)."""

    def __init__(self, image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32,
                 norm_eps=1e-5):
        super().__init__()
        assert image_size % patch_size == 0 and dim % num_heads == 0
        self.image_size, self.patch_size, self.dim, self.num_heads, self.num_layers = image_size, patch_size, dim, num_heads, num_layers
        self.num_patches, self.norm_eps = (image_size // patch_size) ** 2, norm_eps
        gain = dim ** -0.5
        self.patch_embedding = nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.cls_embedding = nn.Parameter(gain * torch.randn(1, 1, dim))
        self.pos_embedding = nn.Parameter(gain * torch.randn(1, self.num_patches + 1, dim))
        self.pre_norm = nn.LayerNorm(dim, eps=norm_eps)
        self.transformer = nn.Sequential(*[_AttentionBlock(dim, mlp_ratio, num_heads, norm_eps) for _ in range(num_layers)])
        self.post_norm = nn.LayerNorm(dim, eps=norm_eps)              # not evaluated on the use_31_block path
        self.head = nn.Parameter(gain * torch.randn(dim, out_dim))    # idem
        self._packed = _Packed()

    def _lin(self, lin: nn.Linear, key):
        return self._packed.get(key, (lin.weight, lin.bias), lambda: (_bf16(lin.weight), lin.bias.detach().float().contiguous()))

    @torch.no_grad()
    def forward(self, x: torch.Tensor, interpolation: bool = False, use_31_block: bool = True) -> torch.Tensor:
        """x fp32 [B, 3, H, W], already resized and normalised -> fp32 [B, 1 + (H/p)^2, dim]."""
        if interpolation:
            raise NotImplementedError("position-embedding interpolation (clip.py:22-39): CLIPModel.visual resizes to 224")
        dev = self.cls_embedding.device
        x = x.to(dev, torch.float32).contiguous()
        B, _, Hh, Ww = x.shape
        p, d, H = self.patch_size, self.dim, self.num_heads
        Dh = d // H
        n = (Hh // p) * (Ww // p)
        assert n == self.num_patches, "image size does not match the position embedding"
        kin = 3 * p * p
        Kp = _round_up(kin, 8)
        wpe = self._packed.get("pe", (self.patch_embedding.weight,), lambda: _pad_cols(
            self.patch_embedding.weight.detach().float().reshape(d, kin), Kp))
        tok = torch.empty(B, n, d, dtype=torch.float32, device=dev)
        for b in range(B):                                             # Conv2d(k = s = p) = GEMM over the patches
            cols = ops.patchify(x[b][:, None].contiguous(), (1, p, p), Kp)         # [n, Kp] bf16, column (c, i, j)
            ops.gemm(cols, wpe, out=tok[b], epilogue=EPI_F32)
        t = ops.vit_embed(tok, self.cls_embedding.detach().float().reshape(d).contiguous(),
                          self.pos_embedding.detach().float().reshape(n + 1, d).contiguous())
        t = ops.layernorm_f32(t, self.pre_norm.weight.detach().float(), self.pre_norm.bias.detach().float(), self.norm_eps)
        hi = self.num_layers - 1 if use_31_block else self.num_layers
        for b in range(B):
            self.run_layers(t[b], 0, hi)                               # fp32 [L, d] view, updated in place
        return t

    @torch.no_grad()
    def run_layers(self, xb: torch.Tensor, lo: int = 0, hi: Optional[int] = None) -> torch.Tensor:
        """Blocks [lo, hi) of clip.py:130-157 (pre-norm attention + erf-GELU MLP) IN PLACE on one sample's fp32 stream
        [1 + patches, dim]."""
        dev = xb.device
        L, d = xb.shape
        H = self.num_heads
        Dh = d // H
        Lp = _round_up(L, 8)
        blocks = list(self.transformer)
        for li in range(lo, len(blocks) if hi is None else hi):
            blk = blocks[li]
            h = ops.layernorm_modulate(xb, self.norm_eps, 0.0, mul0=blk.norm1.weight.detach().float(),
                                       add0=blk.norm1.bias.detach().float())
            wqkv, bqkv = self._lin(blk.attn.to_qkv, (li, "qkv"))
            qk = ops.gemm(h, wqkv[:2 * d], bias=bqkv[:2 * d].contiguous())     # [L, 2d]: q | k (view(b,s,3,n,d), clip.py:78)
            q, k = qk[:, :d].contiguous(), qk[:, d:].contiguous()
            vt = torch.zeros(d, Lp, dtype=torch.bfloat16, device=dev)
            ops.gemm_raw(ptr(wqkv, 2 * d * d), ptr(h), ptr(vt), d, L, d, d, d, Lp, EPI_BF16,
                         bias=ptr(bqkv, 2 * d), bias_mode=ops.BIAS_M)
            o = encoder_attention(q, k, vt, H, Dh, L, Dh ** -0.5)
            wp, bp = self._lin(blk.attn.proj, (li, "proj"))
            ops.gemm_raw(ptr(o), ptr(wp), ptr(xb), L, d, d, d, d, d, EPI_RESID, bias=ptr(bp), bias_mode=ops.BIAS_N,
                         gate_const=1.0)
            h = ops.layernorm_modulate(xb, self.norm_eps, 0.0, mul0=blk.norm2.weight.detach().float(),
                                       add0=blk.norm2.bias.detach().float())
            w1, b1 = self._lin(blk.mlp[0], (li, "fc1"))
            w2, b2 = self._lin(blk.mlp[2], (li, "fc2"))
            u = ops.gemm(h, w1, bias=b1, epilogue=EPI_GELU_ERF)
            m = w1.shape[0]
            ops.gemm_raw(ptr(u), ptr(w2), ptr(xb), L, d, m, m, m, d, EPI_RESID, bias=ptr(b2), bias_mode=ops.BIAS_N,
                         gate_const=1.0)
        return xb


def _pad_cols(w: torch.Tensor, Kp: int) -> torch.Tensor:
    out = torch.zeros(w.shape[0], Kp, dtype=torch.float32, device=w.device)
    out[:, :w.shape[1]] = w
    return ops.cast_bf16(out)


def clip_xlm_roberta_vit_h_14(device="cpu", **kwargs) -> VisionTransformer:
    """The vision tower of clip.py:468-495."""
    cfg = dict(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32)
    cfg.update(kwargs)
    with torch.device(device):
        return VisionTransformer(**cfg)


class CLIPModel:
    """clip.py:501-542 (vision side)."""

    def __init__(self, dtype=torch.float16, device=None, checkpoint_path=None, tokenizer_path=None,
                 model: Optional[VisionTransformer] = None):
        self.dtype, self.checkpoint_path, self.tokenizer_path = dtype, checkpoint_path, tokenizer_path
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if model is None:
            model = clip_xlm_roberta_vit_h_14(device=self.device)
            if checkpoint_path is not None:
                sd = torch.load(checkpoint_path, map_location="cpu")
                model.load_state_dict({k[len("visual."):]: v.float() for k, v in sd.items() if k.startswith("visual.")})
        self.model = model.eval().requires_grad_(False).to(self.device)

    @torch.no_grad()
    def visual(self, videos: List[torch.Tensor]) -> torch.Tensor:
        """videos: list of [3, T, H, W] in [-1, 1] -> [sum T, 257, 1280] (clip.py:527-542)."""
        size = (self.model.image_size,) * 2
        x = torch.cat([F.interpolate(u.transpose(0, 1).float(), size=size, mode="bicubic", align_corners=False)
                       for u in videos])                               # host-side preprocessing, as the reference's transforms
        mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
        x = (x * 0.5 + 0.5 - mean) / std
        return self.model(x.to(self.device), use_31_block=True)
