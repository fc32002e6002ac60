"""fp32 CPU restatement of the Wan 3D causal VAE (TEST INFRASTRUCTURE).

Restates ``WanVAE_.encode/decode`` (seaweed_apt/wan/modules/vae.py:516-568)
as a *streaming* computation: the video is fed in temporal chunks (encode:
1, 4, 4, ... pixel frames, vae.py:520-534; decode: one latent frame at a time,
vae.py:554-566) and every causal convolution keeps the last two frames of its
own input stream as history (what the reference's ``feat_cache`` slots hold,
vae.py:205-217), zero-filled before the first frame (vae.py:28-36).  Takes a
plain state dict with the reference's key names (``encoder.*, conv1.*,
conv2.*, decoder.*``).

Pinned against the real reference by oracle/make_golden.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F

# latent normalisation tables, vae.py:629-636
LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


@dataclass
class VAEConfig:
    """vae.py:597-604 defaults of _video_vae."""
    dim: int = 96
    z_dim: int = 16
    dim_mult: tuple = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_downsample: tuple = (False, True, True)


# --------------------------------------------------------------------------
# layer plans: the module tree of Encoder3d / Decoder3d as flat op lists
# --------------------------------------------------------------------------
def encoder_plan(cfg: VAEConfig):
    """vae.py:283-316.  Ops: ('conv', key, cin, cout, (kt,kh,kw)) causal conv with
    history; ('res', prefix, cin, cout); ('down', prefix, c, temporal);
    ('attn', prefix, c); ('head', prefix, cin, cout)."""
    dims = [cfg.dim * u for u in (1,) + tuple(cfg.dim_mult)]
    ops = [("conv", "encoder.conv1", 3, dims[0], (3, 3, 3))]
    idx = 0
    out_dim = dims[0]
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            ops.append(("res", f"encoder.downsamples.{idx}", in_dim, out_dim))
            idx += 1
            in_dim = out_dim
        if i != len(cfg.dim_mult) - 1:
            ops.append(("down", f"encoder.downsamples.{idx}", out_dim, cfg.temperal_downsample[i]))
            idx += 1
    ops += [("res", "encoder.middle.0", out_dim, out_dim),
            ("attn", "encoder.middle.1", out_dim),
            ("res", "encoder.middle.2", out_dim, out_dim),
            ("head", "encoder.head", out_dim, cfg.z_dim * 2)]
    return ops


def decoder_plan(cfg: VAEConfig):
    """vae.py:388-421."""
    dm = tuple(cfg.dim_mult)
    dims = [cfg.dim * u for u in (dm[-1],) + dm[::-1]]
    t_up = tuple(cfg.temperal_downsample)[::-1]
    ops = [("conv", "decoder.conv1", cfg.z_dim, dims[0], (3, 3, 3)),
           ("res", "decoder.middle.0", dims[0], dims[0]),
           ("attn", "decoder.middle.1", dims[0]),
           ("res", "decoder.middle.2", dims[0], dims[0])]
    idx = 0
    out_dim = dims[0]
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(cfg.num_res_blocks + 1):
            ops.append(("res", f"decoder.upsamples.{idx}", in_dim, out_dim))
            idx += 1
            in_dim = out_dim
        if i != len(dm) - 1:
            ops.append(("up", f"decoder.upsamples.{idx}", out_dim, t_up[i]))
            idx += 1
    ops.append(("head", "decoder.head", out_dim, 3))
    return ops


def param_shapes(cfg: VAEConfig):
    shapes = {}

    def conv(key, cin, cout, k):
        shapes[key + ".weight"] = (cout, cin) + tuple(k)
        shapes[key + ".bias"] = (cout,)

    def walk(ops):
        for op in ops:
            kind, key = op[0], op[1]
            if kind == "conv":
                conv(key, op[2], op[3], op[4])
            elif kind == "res":
                cin, cout = op[2], op[3]
                shapes[key + ".residual.0.gamma"] = (cin, 1, 1, 1)
                conv(key + ".residual.2", cin, cout, (3, 3, 3))
                shapes[key + ".residual.3.gamma"] = (cout, 1, 1, 1)
                conv(key + ".residual.6", cout, cout, (3, 3, 3))
                if cin != cout:
                    conv(key + ".shortcut", cin, cout, (1, 1, 1))
            elif kind == "down":
                c = op[2]
                conv(key + ".resample.1", c, c, (3, 3))
                if op[3]:
                    conv(key + ".time_conv", c, c, (3, 1, 1))
            elif kind == "up":
                c = op[2]
                conv(key + ".resample.1", c, c // 2, (3, 3))
                if op[3]:
                    conv(key + ".time_conv", c, 2 * c, (3, 1, 1))
            elif kind == "attn":
                c = op[2]
                shapes[key + ".norm.gamma"] = (c, 1, 1)
                conv(key + ".to_qkv", c, 3 * c, (1, 1))
                conv(key + ".proj", c, c, (1, 1))
            elif kind == "head":
                shapes[key + ".0.gamma"] = (op[2], 1, 1, 1)
                conv(key + ".2", op[2], op[3], (3, 3, 3))

    walk(encoder_plan(cfg))
    conv("conv1", cfg.z_dim * 2, cfg.z_dim * 2, (1, 1, 1))
    conv("conv2", cfg.z_dim, cfg.z_dim, (1, 1, 1))
    walk(decoder_plan(cfg))
    return shapes


def synth_state_dict(cfg: VAEConfig, tag: str = "vae"):
    """Deterministic synthetic VAE weights: variance-preserving conv init so a
    random-weight decoder produces O(1) activations through ~30 layers."""
    from . import detgen
    sd = {}
    for name, shape in param_shapes(cfg).items():
        key = f"{tag}/{name}"
        if name.endswith("gamma"):
            v = 1.0 + detgen.uniform(key, shape, -0.2, 0.2)
        elif name.endswith("bias"):
            v = detgen.uniform(key, shape, -0.05, 0.05)
        else:
            fan_in = int(math.prod(shape[1:]))
            a = math.sqrt(3.0 / fan_in)
            v = detgen.uniform(key, shape, -a, a)
        sd[name] = torch.from_numpy(v.copy())
    return sd


# --------------------------------------------------------------------------
# streaming executor
# --------------------------------------------------------------------------
def _rms_silu(x, gamma):
    """vae.py:39-54 (channel-wise L2 normalise * sqrt(C) * gamma) then SiLU."""
    c = x.shape[1]
    return F.silu(F.normalize(x, dim=1) * (c ** 0.5) * gamma.view(1, c, *([1] * (x.dim() - 2))))


class _Stream:
    """History of every causal conv of one encode/decode call (the reference's
    ``_feat_map`` / ``_enc_feat_map`` lists, vae.py:582-589), keyed by name."""

    def __init__(self):
        self.hist = {}
        self.seen = set()


def _causal_conv(sd, key, x, st: _Stream, stride_t: int = 1):
    """vae.py:17-36 + the cache bookkeeping of vae.py:205-217: convolve over
    [history (<=2 frames, zero-filled at stream start) | x] in time, 'same'
    zero padding in space; then remember the last two input frames."""
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        h = st.hist.get(key)
        need = kt - 1
        if h is None:
            h = x.new_zeros(x.shape[0], x.shape[1], need, *x.shape[3:])
        elif h.shape[2] < need:
            h = torch.cat([h.new_zeros(h.shape[0], h.shape[1], need - h.shape[2], *h.shape[3:]), h], 2)
        xin = torch.cat([h[:, :, -need:], x], dim=2)
        allf = torch.cat([st.hist[key], x], 2) if key in st.hist else x
        st.hist[key] = allf[:, :, -2:].clone()
    else:
        xin = x
    xin = F.pad(xin, (kw // 2, kw // 2, kh // 2, kh // 2))
    return F.conv3d(xin, w, b, stride=(stride_t, 1, 1))


def _res_block(sd, key, cin, cout, x, st):
    """vae.py:186-220."""
    h = F.conv3d(x, sd[key + ".shortcut.weight"], sd[key + ".shortcut.bias"]) if cin != cout else x
    y = _rms_silu(x, sd[key + ".residual.0.gamma"])
    y = _causal_conv(sd, key + ".residual.2", y, st)
    y = _rms_silu(y, sd[key + ".residual.3.gamma"])
    y = _causal_conv(sd, key + ".residual.6", y, st)
    return y + h


def _mid_attention(sd, key, x):
    """vae.py:223-262 — per-frame single-head attention over H*W tokens."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.normalize(y, dim=1) * (c ** 0.5) * sd[key + ".norm.gamma"].view(1, c, 1, 1)
    qkv = F.conv2d(y, sd[key + ".to_qkv.weight"], sd[key + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 3, c, h * w).permute(1, 0, 3, 2)      # each [bt, hw, c]
    p = torch.softmax(torch.matmul(q, k.transpose(1, 2)) / math.sqrt(c), dim=-1)
    o = torch.matmul(p, v).permute(0, 2, 1).reshape(b * t, c, h, w)
    o = F.conv2d(o, sd[key + ".proj.weight"], sd[key + ".proj.bias"])
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def _per_frame(fn, x):
    b, c, t, h, w = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def _upsample(sd, key, c, temporal, x, st):
    """vae.py:101-141 (upsample2d / upsample3d).  The first chunk of a stream
    bypasses the temporal up-conv entirely (the reference's 'Rep' marker);
    the up-conv's own input stream therefore starts, with zero history, at
    the second chunk."""
    if temporal:
        if key not in st.seen:
            st.seen.add(key)
        else:
            b, _, t, h, w = x.shape
            y = _causal_conv(sd, key + ".time_conv", x, st)          # [b, 2c, t, h, w]
            y = y.reshape(b, 2, c, t, h, w)
            x = torch.stack((y[:, 0], y[:, 1]), dim=3).reshape(b, c, 2 * t, h, w)
    def f(u):
        u = F.interpolate(u, scale_factor=2.0, mode="nearest-exact")
        return F.conv2d(u, sd[key + ".resample.1.weight"], sd[key + ".resample.1.bias"], padding=1)
    return _per_frame(f, x)


def _downsample(sd, key, c, temporal, x, st):
    """vae.py:87-96,138-160 (downsample2d / downsample3d).  Spatial: zero-pad
    right/bottom by one, 3x3 stride-2 conv.  Temporal: first chunk passes
    through; later chunks apply a (3,1,1) stride-2 conv over [last frame of
    the previous chunk | chunk]."""
    def f(u):
        return F.conv2d(F.pad(u, (0, 1, 0, 1)), sd[key + ".resample.1.weight"],
                        sd[key + ".resample.1.bias"], stride=2)
    x = _per_frame(f, x)
    if temporal:
        tk = key + ".time_conv"
        if tk not in st.hist:
            st.hist[tk] = x[:, :, -1:].clone()
        else:
            xin = torch.cat([st.hist[tk], x], dim=2)
            st.hist[tk] = x[:, :, -1:].clone()
            x = F.conv3d(xin, sd[tk + ".weight"], sd[tk + ".bias"], stride=(2, 1, 1))
    return x


def _run_plan(sd, ops, x, st):
    for op in ops:
        kind, key = op[0], op[1]
        if kind == "conv":
            x = _causal_conv(sd, key, x, st)
        elif kind == "res":
            x = _res_block(sd, key, op[2], op[3], x, st)
        elif kind == "attn":
            x = _mid_attention(sd, key, x)
        elif kind == "down":
            x = _downsample(sd, key, op[2], op[3], x, st)
        elif kind == "up":
            x = _upsample(sd, key, op[2], op[3], x, st)
        elif kind == "head":
            x = _causal_conv(sd, key + ".2", _rms_silu(x, sd[key + ".0.gamma"]), st)
    return x


def _scale(cfg):
    assert cfg.z_dim == 16
    mean = torch.tensor(LATENT_MEAN, dtype=torch.float32)
    inv_std = 1.0 / torch.tensor(LATENT_STD, dtype=torch.float32)
    return mean, inv_std


@torch.no_grad()
def vae_encode(sd, cfg: VAEConfig, video: torch.Tensor, normalise: bool = True,
               max_chunks: int = None) -> torch.Tensor:
    """WanVAE.encode for one clip (vae.py:647-655, :516-542).
    video [3, T, H, W] (T = 4n+1) -> latent mean [z, (T-1)/4+1, H/8, W/8]."""
    x = video.unsqueeze(0).float()
    ops = encoder_plan(cfg)
    st = _Stream()
    n_chunks = 1 + (x.shape[2] - 1) // 4
    if max_chunks is not None:
        n_chunks = min(n_chunks, max_chunks)
    outs = []
    for i in range(n_chunks):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(_run_plan(sd, ops, chunk, st))
    out = torch.cat(outs, dim=2)
    mu = F.conv3d(out, sd["conv1.weight"], sd["conv1.bias"])[:, :cfg.z_dim]
    if normalise:
        mean, inv_std = _scale(cfg)
        mu = (mu - mean.view(1, -1, 1, 1, 1)) * inv_std.view(1, -1, 1, 1, 1)
    return mu.squeeze(0).float()


@torch.no_grad()
def vae_decode(sd, cfg: VAEConfig, z: torch.Tensor, normalise: bool = True,
               max_chunks: int = None, clamp: bool = True) -> torch.Tensor:
    """WanVAE.decode for one clip (vae.py:657-663, :544-568).
    z [z, T', h, w] -> video [3, 4(T'-1)+1, 8h, 8w] clamped to [-1, 1]."""
    z = z.unsqueeze(0).float()
    if normalise:
        mean, inv_std = _scale(cfg)
        z = z / inv_std.view(1, -1, 1, 1, 1) + mean.view(1, -1, 1, 1, 1)
    x = F.conv3d(z, sd["conv2.weight"], sd["conv2.bias"])
    ops = decoder_plan(cfg)
    st = _Stream()
    n = x.shape[2] if max_chunks is None else min(x.shape[2], max_chunks)
    outs = [_run_plan(sd, ops, x[:, :, i:i + 1], st) for i in range(n)]
    out = torch.cat(outs, dim=2).float()
    if clamp:
        out = out.clamp_(-1, 1)
    return out.squeeze(0)
