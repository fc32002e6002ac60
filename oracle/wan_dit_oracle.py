"""fp32 CPU restatement of the Wan2.1 DiT forward (TEST INFRASTRUCTURE).

Restates, in plain functional PyTorch on CPU tensors, what the reference's
``WanModel.forward`` computes (seaweed_apt/wan/modules/model.py:502-563) when
it runs in fp32 — which is what the reference itself does on CPU, where its
``torch.cuda.amp.autocast`` contexts are inert.  Every function cites the
reference lines it follows.  It takes a plain ``state_dict`` with the
reference's key names so the same weights drive the reference,
this oracle and the HIP product path.

Pinned against the real reference by oracle/make_golden.py (vectors under
tests/golden/).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    """Constructor arguments of the reference WanModel (model.py:387-404)."""
    model_type: str = "t2v"
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    text_len: int = 512
    in_dim: int = 16
    dim: int = 2048
    ffn_dim: int = 8192
    freq_dim: int = 256
    text_dim: int = 4096
    out_dim: int = 16
    num_heads: int = 16
    num_layers: int = 32
    qk_norm: bool = True
    cross_attn_norm: bool = True
    eps: float = 1e-6

    @staticmethod
    def wan_t2v_1_3b() -> "DiTConfig":
        # seaweed_apt/wan/configs/wan_t2v_1_3B.py:20-29
        return DiTConfig(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)

    @staticmethod
    def wan_i2v_14b() -> "DiTConfig":
        # seaweed_apt/wan/configs/wan_i2v_14B.py:26-35
        return DiTConfig(model_type="i2v", in_dim=36, dim=5120, ffn_dim=13824,
                         num_heads=40, num_layers=40)


# --------------------------------------------------------------------------
# parameter shapes (the state-dict contract)
# --------------------------------------------------------------------------
def param_shapes(cfg: DiTConfig) -> "dict[str, tuple]":
    d, f = cfg.dim, cfg.ffn_dim
    pt, ph, pw = cfg.patch_size
    shapes = {
        "patch_embedding.weight": (d, cfg.in_dim, pt, ph, pw),
        "patch_embedding.bias": (d,),
        "text_embedding.0.weight": (d, cfg.text_dim),
        "text_embedding.0.bias": (d,),
        "text_embedding.2.weight": (d, d),
        "text_embedding.2.bias": (d,),
        "time_embedding.0.weight": (d, cfg.freq_dim),
        "time_embedding.0.bias": (d,),
        "time_embedding.2.weight": (d, d),
        "time_embedding.2.bias": (d,),
        "time_projection.1.weight": (6 * d, d),
        "time_projection.1.bias": (6 * d,),
    }
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        shapes[p + "modulation"] = (1, 6, d)
        for attn in ("self_attn", "cross_attn"):
            for lin in ("q", "k", "v", "o"):
                shapes[p + f"{attn}.{lin}.weight"] = (d, d)
                shapes[p + f"{attn}.{lin}.bias"] = (d,)
            if cfg.qk_norm:
                shapes[p + f"{attn}.norm_q.weight"] = (d,)
                shapes[p + f"{attn}.norm_k.weight"] = (d,)
        if cfg.model_type == "i2v":
            shapes[p + "cross_attn.k_img.weight"] = (d, d)
            shapes[p + "cross_attn.k_img.bias"] = (d,)
            shapes[p + "cross_attn.v_img.weight"] = (d, d)
            shapes[p + "cross_attn.v_img.bias"] = (d,)
            if cfg.qk_norm:
                shapes[p + "cross_attn.norm_k_img.weight"] = (d,)
        if cfg.cross_attn_norm:
            shapes[p + "norm3.weight"] = (d,)
            shapes[p + "norm3.bias"] = (d,)
        shapes[p + "ffn.0.weight"] = (f, d)
        shapes[p + "ffn.0.bias"] = (f,)
        shapes[p + "ffn.2.weight"] = (d, f)
        shapes[p + "ffn.2.bias"] = (d,)
    shapes["head.modulation"] = (1, 2, d)
    shapes["head.head.weight"] = (math.prod(cfg.patch_size) * cfg.out_dim, d)
    shapes["head.head.bias"] = (math.prod(cfg.patch_size) * cfg.out_dim,)
    if cfg.model_type == "i2v":
        shapes["img_emb.proj.0.weight"] = (1280,)
        shapes["img_emb.proj.0.bias"] = (1280,)
        shapes["img_emb.proj.1.weight"] = (1280, 1280)
        shapes["img_emb.proj.1.bias"] = (1280,)
        shapes["img_emb.proj.3.weight"] = (d, 1280)
        shapes["img_emb.proj.3.bias"] = (d,)
        shapes["img_emb.proj.4.weight"] = (d,)
        shapes["img_emb.proj.4.bias"] = (d,)
    return shapes


def synth_state_dict(cfg: DiTConfig, tag: str = "dit") -> "dict[str, torch.Tensor]":
    """Deterministic synthetic weights (oracle/detgen.py) shaped like the
    reference's init_weights (model.py:590-612) but with non-degenerate biases,
    norm gains and head weight (the reference zero-inits head.head.weight,
    which would make every output equal to the bias)."""
    from . import detgen
    sd = {}
    for name, shape in param_shapes(cfg).items():
        key = f"{tag}/{name}"
        if name.endswith("modulation"):
            v = detgen.normalish(key, shape, std=1.0 / math.sqrt(cfg.dim))
        elif "norm" in name and name.endswith("weight") or name in (
                "img_emb.proj.0.weight", "img_emb.proj.4.weight"):
            v = 1.0 + detgen.uniform(key, shape, -0.2, 0.2)
        elif name.endswith("bias"):
            v = detgen.uniform(key, shape, -0.05, 0.05)
        elif name.startswith(("text_embedding", "time_embedding")):
            v = detgen.normalish(key, shape, std=0.02)
        else:
            v = detgen.xavier(key, shape)
        sd[name] = torch.from_numpy(v.copy())
    return sd


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """model.py:17-27 — fp64 [cos | sin] table of t * 10000^(-i/half)."""
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half)
    ang = torch.outer(pos, inv)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=1)


def rope_table(head_dim: int, max_len: int = 1024, theta: float = 10000.0):
    """model.py:31-38,485-492 — per-axis rotary angles, fp64.

    Returns angles [max_len, head_dim/2]; the first c-2*(c//3) columns index
    the frame axis, then c//3 for height, c//3 for width (c = head_dim/2),
    each built from its own sub-dimension (d-4*(d//6), 2*(d//6), 2*(d//6))."""
    d = head_dim
    parts = []
    for sub in (d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6)):
        inv = 1.0 / torch.pow(theta, torch.arange(0, sub, 2, dtype=torch.float64) / sub)
        parts.append(torch.outer(torch.arange(max_len, dtype=torch.float64), inv))
    return torch.cat(parts, dim=1)


def rope_apply(x: torch.Tensor, grid_sizes, angles: torch.Tensor) -> torch.Tensor:
    """model.py:42-69 — rotate consecutive (even, odd) channel pairs of every
    head by the angle of the token's (f, h, w) grid position; tokens beyond
    f*h*w are left untouched.  x: [B, S, N, D]; math in fp64, result fp32."""
    B, S, N, D = x.shape
    c = D // 2
    cf, ch, cw = c - 2 * (c // 3), c // 3, c // 3
    af, ah, aw = angles.split([cf, ch, cw], dim=1)
    out = []
    for b, (f, h, w) in enumerate(grid_sizes):
        n_tok = f * h * w
        ang = torch.cat([
            af[:f].view(f, 1, 1, cf).expand(f, h, w, cf),
            ah[:h].view(1, h, 1, ch).expand(f, h, w, ch),
            aw[:w].view(1, 1, w, cw).expand(f, h, w, cw),
        ], dim=-1).reshape(n_tok, 1, c)
        xb = x[b, :n_tok].to(torch.float64).reshape(n_tok, N, c, 2)
        cos, sin = torch.cos(ang), torch.sin(ang)
        re = xb[..., 0] * cos - xb[..., 1] * sin
        im = xb[..., 0] * sin + xb[..., 1] * cos
        rot = torch.stack([re, im], dim=-1).reshape(n_tok, N, D)
        out.append(torch.cat([rot.to(x.dtype), x[b, n_tok:]], dim=0))
    return torch.stack(out).float()


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """model.py:72-88 — RMS over the full model dim, learned gain."""
    xf = x.float()
    return xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def layer_norm(x, eps, weight=None, bias=None):
    """model.py:91-104 — LayerNorm in fp32 (affine only for norm3)."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps)


def masked_attention(q, k, v, k_lens=None, q_chunk: int = 2048):
    """attention.py:24-130 restated: softmax(q k^T / sqrt(D)) v per head,
    non-causal, with keys at index >= k_lens[b] excluded (what the varlen
    packing at attention.py:79-80 achieves).  q: [B, Lq, N, D], k/v: [B, Lk, N, D].
    Processed in query chunks so S=32760 fits in host memory."""
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    scale = 1.0 / math.sqrt(D)
    out = torch.empty(B, Lq, N, v.shape[-1], dtype=torch.float32)
    for b in range(B):
        kl = Lk if k_lens is None else int(k_lens[b])
        if kl == 0:
            out[b].zero_()
            continue
        kb = k[b, :kl].float().permute(1, 2, 0)      # [N, D, kl]
        vb = v[b, :kl].float().permute(1, 0, 2)      # [N, kl, Dv]
        for s0 in range(0, Lq, q_chunk):
            qb = q[b, s0:s0 + q_chunk].float().permute(1, 0, 2)   # [N, c, D]
            p = torch.softmax(torch.matmul(qb, kb) * scale, dim=-1)
            out[b, s0:s0 + q_chunk] = torch.matmul(p, vb).permute(1, 0, 2)
    return out


def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def self_attention(sd, p, cfg, x, seq_lens, grid_sizes, angles):
    """model.py:132-161."""
    B, S, _ = x.shape
    N, D = cfg.num_heads, cfg.dim // cfg.num_heads
    q, k, v = _lin(sd, p + "q", x), _lin(sd, p + "k", x), _lin(sd, p + "v", x)
    if cfg.qk_norm:
        q = rms_norm(q, sd[p + "norm_q.weight"], cfg.eps)
        k = rms_norm(k, sd[p + "norm_k.weight"], cfg.eps)
    q = rope_apply(q.view(B, S, N, D), grid_sizes, angles)
    k = rope_apply(k.view(B, S, N, D), grid_sizes, angles)
    o = masked_attention(q, k, v.view(B, S, N, D), k_lens=seq_lens)
    return _lin(sd, p + "o", o.flatten(2))


def cross_attention(sd, p, cfg, x, context, context_lens):
    """model.py:164-186 (t2v) and :189-230 (i2v: extra attention over the
    first 257 image tokens, summed before the output projection)."""
    B = x.shape[0]
    N, D = cfg.num_heads, cfg.dim // cfg.num_heads
    q = _lin(sd, p + "q", x)
    if cfg.qk_norm:
        q = rms_norm(q, sd[p + "norm_q.weight"], cfg.eps)
    q = q.view(B, -1, N, D)
    img_o = None
    if cfg.model_type == "i2v":
        ctx_img, context = context[:, :257], context[:, 257:]
        k_img = _lin(sd, p + "k_img", ctx_img)
        if cfg.qk_norm:
            k_img = rms_norm(k_img, sd[p + "norm_k_img.weight"], cfg.eps)
        v_img = _lin(sd, p + "v_img", ctx_img)
        img_o = masked_attention(q, k_img.view(B, -1, N, D), v_img.view(B, -1, N, D), None)
    k = _lin(sd, p + "k", context)
    if cfg.qk_norm:
        k = rms_norm(k, sd[p + "norm_k.weight"], cfg.eps)
    v = _lin(sd, p + "v", context)
    o = masked_attention(q, k.view(B, -1, N, D), v.view(B, -1, N, D), context_lens)
    o = o.flatten(2)
    if img_o is not None:
        o = o + img_o.flatten(2)
    return _lin(sd, p + "o", o)


def attention_block(sd, i, cfg, x, e0, seq_lens, grid_sizes, angles, context, context_lens, freeze_ffn=False):
    """model.py:279-330 (WanAttentionBlock.forward + cross_attn_ffn).  The
    reference's block_idx>10 CPU-offloaded FFN (model.py:317-324) is, in
    inference, the same fp32 FFN; only its gradient differs."""
    p = f"blocks.{i}."
    e = (sd[p + "modulation"] + e0).chunk(6, dim=1)            # 6 x [B,1,d]
    h = layer_norm(x, cfg.eps) * (1 + e[1]) + e[0]
    y = self_attention(sd, p + "self_attn.", cfg, h, seq_lens, grid_sizes, angles)
    x = x + y * e[2]
    if cfg.cross_attn_norm:
        h = layer_norm(x, cfg.eps, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    else:
        h = x
    x = x + cross_attention(sd, p + "cross_attn.", cfg, h, context, context_lens)
    h = layer_norm(x, cfg.eps) * (1 + e[4]) + e[3]
    if freeze_ffn:
        # model.py:317-324: FFN evaluated under no_grad (on the CPU), then "+ 0 * ffn_input"
        with torch.no_grad():
            y = _lin(sd, p + "ffn.2", F.gelu(_lin(sd, p + "ffn.0", h), approximate="tanh"))
        y = y + 0 * h
    else:
        y = _lin(sd, p + "ffn.2", F.gelu(_lin(sd, p + "ffn.0", h), approximate="tanh"))
    return x + y * e[5]


def embed_inputs(sd, cfg: DiTConfig, x_list, t, context_list, seq_len, clip_fea=None, y=None, extra_tokens=None):
    """model.py:511-537 — everything before the block loop.  ``extra_tokens`` ([B, Ne, dim], already in model
    width) are the OmniHuman condition tokens (oracle/omnihuman_oracle.py): prepended to the embedded text
    context the way the i2v CLIP tokens are (model.py:534-537), so the cross-attention attends to them."""
    if y is not None:
        x_list = [torch.cat([u, v], dim=0) for u, v in zip(x_list, y)]
    emb = [F.conv3d(u.unsqueeze(0).float(), sd["patch_embedding.weight"],
                    sd["patch_embedding.bias"], stride=cfg.patch_size) for u in x_list]
    grid_sizes = [tuple(u.shape[2:]) for u in emb]
    tok = [u.flatten(2).transpose(1, 2) for u in emb]
    seq_lens = torch.tensor([u.shape[1] for u in tok], dtype=torch.long)
    assert int(seq_lens.max()) <= seq_len
    x = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.shape[1], u.shape[2])], dim=1)
                   for u in tok])
    sin = sinusoidal_embedding_1d(cfg.freq_dim, t.reshape(-1)).float()
    e = _lin(sd, "time_embedding.2", F.silu(_lin(sd, "time_embedding.0", sin)))
    e0 = _lin(sd, "time_projection.1", F.silu(e)).unflatten(1, (6, cfg.dim))
    context_lens = torch.tensor([u.shape[0] for u in context_list], dtype=torch.long)
    ctx = torch.stack([torch.cat([u.float(), u.new_zeros(cfg.text_len - u.shape[0], u.shape[1]).float()])
                       for u in context_list])
    ctx = _lin(sd, "text_embedding.2", F.gelu(_lin(sd, "text_embedding.0", ctx), approximate="tanh"))
    if clip_fea is not None:
        c = F.layer_norm(clip_fea.float(), (1280,), sd["img_emb.proj.0.weight"], sd["img_emb.proj.0.bias"])
        c = F.gelu(_lin(sd, "img_emb.proj.1", c))
        c = _lin(sd, "img_emb.proj.3", c)
        c = F.layer_norm(c, (cfg.dim,), sd["img_emb.proj.4.weight"], sd["img_emb.proj.4.bias"])
        ctx = torch.cat([c, ctx], dim=1)
        context_lens = context_lens + c.shape[1]
    if extra_tokens is not None:
        assert cfg.model_type == "t2v" and extra_tokens.shape[0] == ctx.shape[0] and extra_tokens.shape[2] == cfg.dim
        ctx = torch.cat([extra_tokens.float(), ctx], dim=1)
        context_lens = context_lens + extra_tokens.shape[1]
    return x, e, e0, ctx, context_lens, seq_lens, grid_sizes


def head_unpatchify(sd, cfg: DiTConfig, x, e, grid_sizes):
    """model.py:349-359 (Head) and :565-588 (unpatchify)."""
    em = (sd["head.modulation"] + e.unsqueeze(1)).chunk(2, dim=1)
    h = _lin(sd, "head.head", layer_norm(x, cfg.eps) * (1 + em[1]) + em[0])
    outs = []
    for u, g in zip(h, grid_sizes):
        u = u[:math.prod(g)].view(*g, *cfg.patch_size, cfg.out_dim)
        u = torch.einsum("fhwpqrc->cfphqwr", u)
        outs.append(u.reshape(cfg.out_dim, *[a * b for a, b in zip(g, cfg.patch_size)]).float())
    return outs


@torch.no_grad()
def dit_forward(sd, cfg: DiTConfig, x_list: Sequence[torch.Tensor], t: torch.Tensor,
                context_list: Sequence[torch.Tensor], seq_len: int,
                clip_fea: Optional[torch.Tensor] = None, y=None,
                num_layers: Optional[int] = None, return_hidden: bool = False, extra_tokens=None):
    """model.py:502-563.  ``num_layers`` truncates the block loop (used by the
    bounded cpu_baseline sample); ``return_hidden`` returns the residual
    stream after the last executed block instead of the unpatchified output."""
    x, e, e0, ctx, context_lens, seq_lens, grid_sizes = embed_inputs(
        sd, cfg, x_list, t, context_list, seq_len, clip_fea, y, extra_tokens)
    angles = rope_table(cfg.dim // cfg.num_heads)
    L = cfg.num_layers if num_layers is None else num_layers
    for i in range(L):
        x = attention_block(sd, i, cfg, x, e0, seq_lens, grid_sizes, angles, ctx, context_lens)
    if return_hidden:
        return x
    return head_unpatchify(sd, cfg, x, e, grid_sizes)


def dit_forward_autograd(sd, cfg: DiTConfig, x_list, t, context_list, seq_len, reference_ffn_freeze=True,
                         clip_fea=None, y=None, extra_tokens=None):
    """The same forward with autograd enabled (``sd`` tensors with requires_grad): the oracle of the
    training step (distilled_trainer.py:268-301).  ``reference_ffn_freeze`` reproduces the reference's
    block_idx > 10 FFN quirk (model.py:317-324).  ``clip_fea`` / ``y``: the i2v backbone."""
    x, e, e0, ctx, context_lens, seq_lens, grid_sizes = embed_inputs(sd, cfg, x_list, t, context_list, seq_len,
                                                                     clip_fea, y, extra_tokens)
    angles = rope_table(cfg.dim // cfg.num_heads)
    for i in range(cfg.num_layers):
        x = attention_block(sd, i, cfg, x, e0, seq_lens, grid_sizes, angles, ctx, context_lens,
                            freeze_ffn=reference_ffn_freeze and i > 10)
    return head_unpatchify(sd, cfg, x, e, grid_sizes)


def cfg_velocity(sd, cfg, noise, t, ctx_cond, ctx_uncond, seq_len, scale):
    """generate.py:227-229 / text2video.py:238-244 — classifier-free-guided
    velocity  v = u + s (c - u)  from two forwards on the same latent."""
    c = dit_forward(sd, cfg, [noise], t, [ctx_cond], seq_len)[0]
    u = dit_forward(sd, cfg, [noise], t, [ctx_uncond], seq_len)[0]
    return u + scale * (c - u)
