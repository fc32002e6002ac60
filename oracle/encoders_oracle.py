"""CPU oracle (TEST INFRASTRUCTURE — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import
this) of the two prompt-side encoders of the Wan pipelines: the umT5 text encoder
(``seaweed_apt/wan/modules/t5.py``) and the vision tower of the XLM-Roberta CLIP (``seaweed_apt/wan/modules/clip.py``).
fp32 torch, function by function with reference file:line.  State dicts carry the reference's keys.

Pinning: ``oracle/make_golden.py encoders`` imports the real ``T5Encoder`` / ``VisionTransformer`` in place
(``oracle/ref_import.py:load_reference_encoders``) at a small width and writes their outputs to
``tests/golden/encoders_t5_clip.npz``; ``tests/test_oracle_golden.py`` compares this restatement with them.

The reference's LOCAL MODIFICATION of the T5 block is kept (t5.py:166-176): ``T5SelfAttention.forward`` was cut
down to ``x = norm1(x); x = x + attn(x); return x`` — the residual is taken from the NORMALISED stream and the gated
feed-forward (norm2, ffn) never runs.  ``reference_block_quirk=False`` gives the upstream umT5 block
(``x += attn(norm1 x); x += ffn(norm2 x)``) that the weights were trained with.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import detgen


# ----------------------------------------------------------------------------------------------- umT5 encoder
@dataclass
class T5Config:
    vocab: int = 256384
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    shared_pos: bool = False            # umt5_xxl, t5.py:466-479

    @staticmethod
    def tiny():
        return T5Config(vocab=200, dim=128, dim_attn=128, dim_ffn=320, num_heads=2, num_layers=3, num_buckets=32)


def t5_state_dict(cfg: T5Config, tag: str):
    """Deterministic synthetic weights with the reference's keys (T5Encoder, t5.py:272-322)."""
    sd = {"token_embedding.weight": torch.from_numpy(detgen.normalish(f"{tag}/emb", (cfg.vocab, cfg.dim)))}

    def lin(name, o, i, scale=None):
        sd[name] = torch.from_numpy(detgen.normalish(f"{tag}/{name}", (o, i))) * (scale or i ** -0.5)

    for l in range(cfg.num_layers):
        p = f"blocks.{l}."
        sd[p + "norm1.weight"] = 1.0 + 0.1 * torch.from_numpy(detgen.normalish(f"{tag}/{p}n1", (cfg.dim,)))
        sd[p + "norm2.weight"] = 1.0 + 0.1 * torch.from_numpy(detgen.normalish(f"{tag}/{p}n2", (cfg.dim,)))
        # T5 does not scale its scores (t5.py:112): q is initialised (and stays) much smaller than k
        # (init_weights, t5.py:36-39).  Synthetic q such that the scores have a standard deviation of ~3.
        lin(p + "attn.q.weight", cfg.dim_attn, cfg.dim, scale=3.0 * (cfg.dim_attn // cfg.num_heads) ** -0.5 * cfg.dim ** -0.5)
        lin(p + "attn.k.weight", cfg.dim_attn, cfg.dim)
        lin(p + "attn.v.weight", cfg.dim_attn, cfg.dim)
        lin(p + "attn.o.weight", cfg.dim, cfg.dim_attn)
        lin(p + "ffn.gate.0.weight", cfg.dim_ffn, cfg.dim)
        lin(p + "ffn.fc1.weight", cfg.dim_ffn, cfg.dim)
        lin(p + "ffn.fc2.weight", cfg.dim, cfg.dim_ffn)
        if not cfg.shared_pos:
            sd[p + "pos_embedding.embedding.weight"] = torch.from_numpy(
                detgen.normalish(f"{tag}/{p}pos", (cfg.num_buckets, cfg.num_heads)))
    if cfg.shared_pos:
        sd["pos_embedding.embedding.weight"] = torch.from_numpy(
            detgen.normalish(f"{tag}/pos", (cfg.num_buckets, cfg.num_heads)))
    sd["norm.weight"] = 1.0 + 0.1 * torch.from_numpy(detgen.normalish(f"{tag}/norm", (cfg.dim,)))
    return {k: v.float().contiguous() for k, v in sd.items()}


def t5_relative_buckets(lq: int, lk: int, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (t5.py:244-268): int64 [lq, lk]."""
    rel = torch.arange(lk)[None, :] - torch.arange(lq)[:, None]
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def t5_layernorm(x, w, eps=1e-6):
    """T5LayerNorm (t5.py:55-69): RMS normalisation, no mean subtraction, gain."""
    return w * (x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps))


def gelu_tanh(x):
    """t5.py:48-52."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3.0))))


def t5_attention(sd, p, x, mask, pos_bias, n_heads):
    """T5Attention.forward (t5.py:88-120): no 1/sqrt(d) scaling, additive position bias, masked keys at finfo.min."""
    B, L, _ = x.shape
    q = F.linear(x, sd[p + "q.weight"]).view(B, L, n_heads, -1)
    k = F.linear(x, sd[p + "k.weight"]).view(B, L, n_heads, -1)
    v = F.linear(x, sd[p + "v.weight"]).view(B, L, n_heads, -1)
    bias = x.new_zeros(B, n_heads, L, L)
    if pos_bias is not None:
        bias = bias + pos_bias
    if mask is not None:
        bias = bias.masked_fill(mask.view(B, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
    attn = F.softmax(attn.float(), dim=-1)
    o = torch.einsum("bnij,bjnc->binc", attn, v).reshape(B, L, -1)
    return F.linear(o, sd[p + "o.weight"])


def t5_encode(sd, cfg: T5Config, ids: torch.Tensor, mask: torch.Tensor = None, reference_block_quirk: bool = True):
    """T5Encoder.forward (t5.py:305-322) in eval mode: ids int64 [B, L], mask [B, L] (1 = token) -> fp32 [B, L, dim]."""
    x = sd["token_embedding.weight"][ids]
    L = ids.shape[1]
    buckets = t5_relative_buckets(L, L, cfg.num_buckets)

    def pos(prefix):
        return sd[prefix + "pos_embedding.embedding.weight"][buckets].permute(2, 0, 1).unsqueeze(0)   # [1, N, L, L]

    shared = pos("") if cfg.shared_pos else None
    for l in range(cfg.num_layers):
        p = f"blocks.{l}."
        e = shared if cfg.shared_pos else pos(p)
        if reference_block_quirk:                                  # t5.py:166-176 as modified in this repository
            x = t5_layernorm(x, sd[p + "norm1.weight"])
            x = x + t5_attention(sd, p + "attn.", x, mask, e, cfg.num_heads)
        else:                                                      # upstream Wan2.1 T5SelfAttention.forward
            x = x + t5_attention(sd, p + "attn.", t5_layernorm(x, sd[p + "norm1.weight"]), mask, e, cfg.num_heads)
            h = t5_layernorm(x, sd[p + "norm2.weight"])
            h = F.linear(h, sd[p + "ffn.fc1.weight"]) * gelu_tanh(F.linear(h, sd[p + "ffn.gate.0.weight"]))
            x = x + F.linear(h, sd[p + "ffn.fc2.weight"])
    return t5_layernorm(x, sd["norm.weight"])


# ----------------------------------------------------------------------------------------------- CLIP vision tower
@dataclass
class ViTConfig:
    image_size: int = 224
    patch_size: int = 14
    dim: int = 1280
    mlp_ratio: int = 4
    num_heads: int = 16
    num_layers: int = 32
    norm_eps: float = 1e-5              # clip_xlm_roberta_vit_h_14, clip.py:468-495

    @staticmethod
    def tiny():
        return ViTConfig(image_size=56, patch_size=14, dim=160, mlp_ratio=4, num_heads=2, num_layers=3)


def vit_state_dict(cfg: ViTConfig, tag: str):
    """Synthetic weights with the keys of the reference VisionTransformer (pool 'token', pre_norm; clip.py:209-273)."""
    d, n = cfg.dim, (cfg.image_size // cfg.patch_size) ** 2 + 1
    g = d ** -0.5

    def nrm(name, shape, scale=1.0):
        return (torch.from_numpy(detgen.normalish(f"{tag}/{name}", shape)) * scale).float().contiguous()

    sd = {"cls_embedding": nrm("cls", (1, 1, d), g), "pos_embedding": nrm("pos", (1, n, d), g),
          "head": nrm("head", (d, 32), g),
          "patch_embedding.weight": nrm("pe", (d, 3, cfg.patch_size, cfg.patch_size), (3 * cfg.patch_size ** 2) ** -0.5),
          "pre_norm.weight": 1.0 + nrm("prw", (d,), 0.1), "pre_norm.bias": nrm("prb", (d,), 0.1),
          "post_norm.weight": 1.0 + nrm("pow", (d,), 0.1), "post_norm.bias": nrm("pob", (d,), 0.1)}
    m = int(d * cfg.mlp_ratio)
    for l in range(cfg.num_layers):
        p = f"transformer.{l}."
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = 1.0 + nrm(p + nm + "w", (d,), 0.1)
            sd[p + nm + ".bias"] = nrm(p + nm + "b", (d,), 0.1)
        sd[p + "attn.to_qkv.weight"], sd[p + "attn.to_qkv.bias"] = nrm(p + "qkv", (3 * d, d), g), nrm(p + "qkvb", (3 * d,), 0.1)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = nrm(p + "proj", (d, d), g), nrm(p + "projb", (d,), 0.1)
        sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"] = nrm(p + "fc1", (m, d), g), nrm(p + "fc1b", (m,), 0.1)
        sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"] = nrm(p + "fc2", (d, m), m ** -0.5), nrm(p + "fc2b", (d,), 0.1)
    return sd


def vit_forward(sd, cfg: ViTConfig, x: torch.Tensor, use_31_block: bool = True):
    """VisionTransformer.forward (clip.py:275-301): x fp32 [B, 3, H, W] (already resized / normalised, CLIPModel.visual
    clip.py:527-542) -> fp32 [B, 1 + (H/p)^2, dim]: all blocks but the last when ``use_31_block``; post_norm and head
    are not applied on that path."""
    B = x.shape[0]
    t = F.conv2d(x, sd["patch_embedding.weight"], stride=cfg.patch_size).flatten(2).permute(0, 2, 1)
    t = torch.cat([sd["cls_embedding"].expand(B, -1, -1), t], dim=1) + sd["pos_embedding"]
    t = F.layer_norm(t, (cfg.dim,), sd["pre_norm.weight"], sd["pre_norm.bias"], cfg.norm_eps)
    n, d = cfg.num_heads, cfg.dim // cfg.num_heads
    for l in range(cfg.num_layers - (1 if use_31_block else 0)):
        p = f"transformer.{l}."
        h = F.layer_norm(t, (cfg.dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.norm_eps)
        q, k, v = F.linear(h, sd[p + "attn.to_qkv.weight"], sd[p + "attn.to_qkv.bias"]).view(B, -1, 3, n, d).unbind(2)
        a = torch.softmax(torch.einsum("binc,bjnc->bnij", q, k) * d ** -0.5, -1)          # flash_attention, clip.py:82
        o = torch.einsum("bnij,bjnc->binc", a, v).reshape(B, -1, cfg.dim)
        t = t + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(t, (cfg.dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.norm_eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))            # activation='gelu' (erf)
        t = t + F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    return t


def clip_preprocess(videos):
    """CLIPModel.visual's preprocessing (clip.py:527-537): each [3, T, H, W] in [-1, 1] -> bicubic 224x224 frames,
    mapped to [0, 1] and normalised with the CLIP mean / std."""
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    x = torch.cat([F.interpolate(u.transpose(0, 1).float(), size=(224, 224), mode="bicubic", align_corners=False)
                   for u in videos])
    return (x * 0.5 + 0.5 - mean) / std
