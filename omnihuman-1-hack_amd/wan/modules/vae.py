"""Wan 3D causal VAE on hand-written gfx950 kernels — drop-in for the
reference's ``wan.modules.vae`` (seaweed_apt/wan/modules/vae.py).

Same module tree / state-dict keys as the reference ``WanVAE_``
(``encoder.*, conv1.*, conv2.*, decoder.*``, vae.py:483-508) so
``Wan2.1_VAE.pth`` loads unchanged, and the same ``WanVAE`` wrapper surface
(``.model``, ``.mean``, ``.std``, ``.scale``, ``encode(list)``, ``decode(list)``,
vae.py:619-663).  The modules only hold parameters; the arithmetic is a
streaming executor over libomh.so kernels (include/omh.h):

  * activations are channels-last bf16 ``[T, H, W, C]``;
  * every causal conv owns an input buffer whose first frames are its temporal
    history (what the reference keeps in ``feat_cache`` and re-concatenates
    each chunk, vae.py:205-217) — the conv kernel never pads in time and the
    producer writes straight behind the history;
  * RMS_norm+SiLU is one HBM pass (``omh_rms_silu_cl``) feeding the conv's
    buffer; the residual add rides in the conv epilogue; the nearest-2x
    upsample is folded into the following conv's addressing; the temporal
    upsample's channel->frame interleave is done by the conv's store;
  * the reference's chunking is kept exactly (encode 1,4,4,... frames,
    decode one latent frame per step, first-chunk bypass of the temporal
    resamplers — vae.py:101-160,516-568) because it defines the numerics.

Compute is bf16 MFMA with fp32 accumulation (the reference runs fp32); the
tolerance is stated in tests/test_gpu_vae.py and DESIGN.md.
"""
import logging
import math

import os

import torch
import torch.nn as nn

from .._backend import ops

__all__ = ["WanVAE"]

CACHE_T = 2


# ----------------------------------------------------------------------------
# parameter containers (names = the reference's state-dict keys)
# ----------------------------------------------------------------------------
class CausalConv3d(nn.Conv3d):
    """vae.py:17-36."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._padding = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self.padding = (0, 0, 0)


class RMS_norm(nn.Module):
    """vae.py:39-54."""

    def __init__(self, dim, channel_first=True, images=True, bias=False):
        super().__init__()
        broadcastable_dims = (1, 1, 1) if not images else (1, 1)
        shape = (dim, *broadcastable_dims) if channel_first else (dim,)
        self.channel_first = channel_first
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(shape))
        self.bias = nn.Parameter(torch.zeros(shape)) if bias else 0.


class Upsample(nn.Upsample):
    pass


class Resample(nn.Module):
    """vae.py:66-99."""

    def __init__(self, dim, mode):
        assert mode in ("none", "upsample2d", "upsample3d", "downsample2d", "downsample3d")
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode == "upsample2d":
            self.resample = nn.Sequential(Upsample(scale_factor=(2., 2.), mode="nearest-exact"),
                                          nn.Conv2d(dim, dim // 2, 3, padding=1))
        elif mode == "upsample3d":
            self.resample = nn.Sequential(Upsample(scale_factor=(2., 2.), mode="nearest-exact"),
                                          nn.Conv2d(dim, dim // 2, 3, padding=1))
            self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode == "downsample2d":
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
        elif mode == "downsample3d":
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            self.resample = nn.Identity()


class ResidualBlock(nn.Module):
    """vae.py:186-200."""

    def __init__(self, in_dim, out_dim, dropout=0.0):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = nn.Sequential(
            RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
            RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout),
            CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()


class AttentionBlock(nn.Module):
    """vae.py:223-238."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)
        nn.init.zeros_(self.proj.weight)


class Encoder3d(nn.Module):
    """vae.py:265-316."""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, dim_mult
        self.num_res_blocks, self.attn_scales, self.temperal_downsample = num_res_blocks, attn_scales, temperal_downsample
        dims = [dim * u for u in [1] + dim_mult]
        scale = 1.0
        self.conv1 = CausalConv3d(3, dims[0], 3, padding=1)
        downsamples = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                downsamples.append(ResidualBlock(in_dim, out_dim, dropout))
                if scale in attn_scales:
                    downsamples.append(AttentionBlock(out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                mode = "downsample3d" if temperal_downsample[i] else "downsample2d"
                downsamples.append(Resample(out_dim, mode=mode))
                scale /= 2.0
        self.downsamples = nn.Sequential(*downsamples)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim, dropout), AttentionBlock(out_dim),
                                    ResidualBlock(out_dim, out_dim, dropout))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(),
                                  CausalConv3d(out_dim, z_dim, 3, padding=1))


class Decoder3d(nn.Module):
    """vae.py:369-421."""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_upsample=[False, True, True], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, dim_mult
        self.num_res_blocks, self.attn_scales, self.temperal_upsample = num_res_blocks, attn_scales, temperal_upsample
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        scale = 1.0 / 2 ** (len(dim_mult) - 2)
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0], dropout), AttentionBlock(dims[0]),
                                    ResidualBlock(dims[0], dims[0], dropout))
        upsamples = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            if i == 1 or i == 2 or i == 3:
                in_dim = in_dim // 2
            for _ in range(num_res_blocks + 1):
                upsamples.append(ResidualBlock(in_dim, out_dim, dropout))
                if scale in attn_scales:
                    upsamples.append(AttentionBlock(out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                mode = "upsample3d" if temperal_upsample[i] else "upsample2d"
                upsamples.append(Resample(out_dim, mode=mode))
                scale *= 2.0
        self.upsamples = nn.Sequential(*upsamples)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 3, 3, padding=1))


def count_conv3d(model):
    return sum(1 for m in model.modules() if isinstance(m, CausalConv3d))


# ----------------------------------------------------------------------------
# streaming executor
# ----------------------------------------------------------------------------
def _round_up(a, b):
    return (a + b - 1) // b * b


class _ConvState:
    """Packed weight + temporal-history input buffer of one convolution."""

    def __init__(self, conv, up2=False, stride_hw=1, pad=None, f32=False):
        w = conv.weight.detach()
        if w.dim() == 4:                      # Conv2d -> [Cout, Cin, 1, kh, kw]
            w = w.unsqueeze(2)
        self.Cout, self.Cin_raw, self.KT, self.KH, self.KW = w.shape
        self.f32 = bool(f32)
        self.pair = False
        cp = _round_up(self.Cin_raw, 8)
        wp = w.float().permute(0, 2, 3, 4, 1)                       # [Cout, kt, kh, kw, Cin]
        if self.f32:
            # fp32-faithful mode: the weight as a bf16 pair per channel, blocks [hi | hi | lo] per tap (pattern 1 of
            # omh_split3_f32) against activations [hi | lo | hi]: the same kernels on a contraction 3x as long —
            # or, where the stream kernel takes the layer (decided at the first slot(), when H x W are known: _layout),
            # pairs interleaved per 16 channels on both operands (omh_conv_args.pair, round 5): 2 C channels
            self._wp = wp.contiguous()
            self.Cin = 3 * cp
            self.w = None
        else:
            self.Cin = cp
            if self.Cin != self.Cin_raw:
                wp = torch.nn.functional.pad(wp, (0, self.Cin - self.Cin_raw))
            self.w = ops.cast_bf16(wp.contiguous().view(self.Cout, -1))
        self.bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        st = conv.stride if isinstance(conv.stride, tuple) else (conv.stride,) * 3
        self.stride_t = st[0] if len(st) == 3 else 1
        self.stride_hw = stride_hw
        self.up2 = up2
        self.pad = (self.KH // 2, self.KW // 2) if pad is None else pad
        self.hist = self.KT - 1 if self.stride_t == 1 else 1      # history frames kept in front of the chunk
        self.buf = None
        self.T = 0
        self.off = 0                                               # frame index of the history inside ``buf``

    # The input buffer is a WINDOW several chunks long that the [history | chunk] region slides through: after a
    # convolution the last ``hist`` frames of the region already ARE the next chunk's history, so the region just
    # moves up by T frames; only when it reaches the end of the buffer are the history frames copied back to the
    # front.  (The first version kept a buffer of exactly hist + T frames and copied the history down after every
    # convolution: 2 frames x 35 layers x 21 chunks — 6.6 % of the VAE's GPU time in `__amd_rocclr_copyBuffer`,
    # profiles/r01_v7_kernel_stats_full_bench.csv.)  Window size: 8 chunks, capped at OMH_VAE_WINDOW_MB (default
    # 1024) per layer — 288 GB of HBM make that an easy trade.
    # fp32-faithful mode: a frame is three channel blocks (480x832 x 288 channels = 230 MB), 1 GB would be a window of
    # [history | one chunk] and the history would be copied back after EVERY convolution (982 copies = 3.6 % of a decode,
    # profiles/r05_vae_fp32_kernel_stats.csv): 8 GB per layer there — 288 GB of HBM make that an easy trade too.
    _WINDOW_BYTES = int(os.environ.get("OMH_VAE_WINDOW_MB", "1024")) << 20
    _WINDOW_BYTES_F32 = int(os.environ.get("OMH_VAE_WINDOW_F32_MB", "8192")) << 20

    def _layout(self, H, W):
        """fp32-faithful mode, first use: the operand layout of this layer — split-bf16 PAIRS (2 C channels) where the
        stream kernel takes the layer, three channel blocks (3 C) everywhere else — and the weight packed to match."""
        taps = self.KT * self.KH * self.KW
        cp = _round_up(self.Cin_raw, 8)
        eff_h, eff_w = (2 * H, 2 * W) if self.up2 else (H, W)
        ok = _PAIR and self.Cin_raw % 16 == 0 and self.stride_t == 1 and self.stride_hw == 1 and \
            ops.conv_pair_supported(self.KT + 1, H, W, 2 * self.Cin_raw, 2, eff_h, eff_w, self.Cout, self.KT, self.KH,
                                    self.KW, pad_h=self.pad[0], pad_w=self.pad[1], up2=self.up2)
        self.pair = bool(ok)
        flat = self._wp.view(self.Cout * taps, self.Cin_raw)
        if self.pair:
            self.Cin = 2 * self.Cin_raw
            self.w = ops.split3(flat, 2, Cp=self.Cin_raw).view(self.Cout, -1)
        else:
            self.Cin = 3 * cp
            self.w = ops.split3(flat, 1, Cp=cp).view(self.Cout, -1)
        self._wp = None

    def slot(self, T, H, W, device):
        """View [T, H, W, Cin] the producer writes the current chunk into."""
        if self.w is None:
            self._layout(H, W)
        need = self.hist + T
        fresh = self.buf is None or tuple(self.buf.shape[1:3]) != (H, W)
        if fresh or self.buf.shape[0] < need:
            old, old_off = (None, 0) if fresh else (self.buf, self.off)
            frame_bytes = H * W * self.Cin * 2
            window = self._WINDOW_BYTES_F32 if self.f32 else self._WINDOW_BYTES
            if self.f32 and torch.device(device).type == "cuda":
                # the 8 GB windows are a trade against FREE memory: next to a DiT, an encoder and
                # training state a layer takes at most 1/32 of what is left (and never less than the 1 GB of the bf16 mode)
                # (what the caching allocator holds but has not handed out counts as available: the windows of the
                # previous decode are exactly that; whole powers of two, so that repeated decodes ask for the same sizes)
                avail = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
                if avail // 32 < window:
                    window = max(self._WINDOW_BYTES, 1 << max(0, (avail // 32).bit_length() - 1))
            cap = need if not self.hist else max(need, min(self.hist + 8 * max(T, 4), window // frame_bytes))
            try:
                self.buf = torch.empty(cap, H, W, self.Cin, dtype=torch.bfloat16, device=device)
            except torch.cuda.OutOfMemoryError:                    # no room for a window: exactly [history | chunk]
                self.buf = torch.empty(need, H, W, self.Cin, dtype=torch.bfloat16, device=device)
            self.off = 0
            if self.hist:
                if old is not None:
                    self.buf[:self.hist].copy_(old[old_off:old_off + self.hist])
                else:
                    self.buf[:self.hist].zero_()                   # the causal zero padding in front of the clip
        elif self.off + need > self.buf.shape[0]:                  # end of the window: history back to the front
            if self.off >= self.hist:
                self.buf[:self.hist].copy_(self.buf[self.off:self.off + self.hist])
            else:                                                  # overlapping ranges: frame by frame, ascending
                for i in range(self.hist):
                    self.buf[i].copy_(self.buf[self.off + i])
            self.off = 0
        self.T = T
        return self.buf[self.off + self.hist:self.off + self.hist + T]

    def set_history(self, frame):
        """Overwrite the (single) history frame — the encoder's first-chunk bypass of a strided time conv."""
        _fill(self.buf[self.off:self.off + 1], frame.unsqueeze(0))

    def run(self, resid=None, out_f32=False, split_n=0, out=None, norm_gamma=None, norm_out=None, norm_only=False):
        """Convolve over [history | chunk]; the last ``hist`` frames of that region become the new history.
        ``norm_gamma`` / ``norm_out`` / ``norm_only``: also the next layer's RMS norm + SiLU (ops.conv_cl)."""
        T, (_, H, W, _) = self.T, self.buf.shape
        x = self.buf[self.off:self.off + self.hist + T]
        if self.stride_t == 1:
            Tout = T
        else:
            Tout = (self.hist + T - self.KT) // self.stride_t + 1
        eff_h, eff_w = (2 * H, 2 * W) if self.up2 else (H, W)
        if self.stride_hw == 1:
            Hout, Wout = eff_h, eff_w
        else:                                               # ZeroPad2d((0,1,0,1)) + 3x3 stride 2 (vae.py:88-90)
            Hout, Wout = (eff_h + 1 - self.KH) // 2 + 1, (eff_w + 1 - self.KW) // 2 + 1
        y = ops.conv_cl(x, self.w, self.bias, Tout, Hout, Wout, self.Cout, self.KT, self.KH, self.KW,
                        stride_t=self.stride_t, stride_hw=self.stride_hw, pad_h=self.pad[0], pad_w=self.pad[1],
                        up2=self.up2, resid=resid, out_f32=out_f32, split_n=split_n, out=out,
                        norm_gamma=norm_gamma, norm_out=norm_out, norm_only=norm_only, pair=self.pair)
        if self.hist:
            self.off += T                                   # the region slides: no copy
        return y


class _Stream:
    """All per-call state of one encode or decode (the reference's _feat_map)."""

    def __init__(self, device, f32=False):
        self.device = device
        self.convs = {}
        self.seen = set()
        # fp32-faithful mode (WanVAE(dtype=torch.float), the reference's default: vae.py:619-624,649-663): every tensor
        # between kernels is fp32, every MFMA operand a bf16 pair (omh_split3_f32) — three products per tile, fp32
        # accumulate; no fused norm epilogues (their output is a single bf16)
        self.f32 = bool(f32)
        self.trunk_f32 = _TRUNK_F32 or self.f32

    def conv(self, key, module, **kw):
        st = self.convs.get(key)
        if st is None:
            st = self.convs[key] = _ConvState(module, f32=self.f32, **kw)
        return st

    def norm(self, x, gamma, out, do_silu=True):
        """RMS norm (+ SiLU) of trunk tensor x into a convolution's input slot."""
        if self.f32:
            return ops.rms_silu_cl_split3(x, gamma, out=out, do_silu=do_silu)
        return ops.rms_silu_cl(x, gamma, out=out, do_silu=do_silu)


def _gamma(norm: RMS_norm):
    return norm.gamma.detach().float().reshape(-1).contiguous()


# The residual trunk (the tensor that runs from block to block: conv1's output, every ResidualBlock / AttentionBlock /
# Resample output) is kept in fp32, as the reference's VAE does throughout (WanVAE(dtype=torch.float), vae.py:619-624):
# RMS statistics are then taken from unrounded values and the skip path is never rounded, so the 2^-9 of a bf16
# rounding per block no longer accumulates over the ~20 blocks of a decode.  Convolution INPUTS are bf16 (the MFMA
# operand type), rounded once from the fp32 trunk.  OMH_VAE_TRUNK=bf16 restores the round-1 executor (A/B, tests).
_TRUNK_F32 = os.environ.get("OMH_VAE_TRUNK", "f32") != "bf16"
_GROUP = max(1, int(os.environ.get("OMH_VAE_GROUP", "5")))      # latent frames per step at the decoder's 2h x 2w stage
# ... and at its 4h x 4w / 8h x 8w stages: 2 latent frames = 8 frames per convolution (3 145 / 6 263 tiles of the stream
# kernel instead of 1 573 / 3 132: the last round of tiles on 256 CUs wastes 5.5 % / 2 % instead of 12 % / 6 %)
_GROUP2 = max(1, int(os.environ.get("OMH_VAE_GROUP2", "2")))


def _fill(slot, x):
    """Write trunk tensor x into a convolution's bf16 input slot (a split-bf16 slot, three channel blocks, in the
    fp32-faithful mode)."""
    if x.dtype == torch.float32 and x.shape[-1] % 16 == 0 and slot.shape[-1] == 2 * x.shape[-1]:
        ops.split3(x.contiguous(), 2, Cp=x.shape[-1], out=slot)             # split-bf16 pairs (omh_conv_args.pair)
    elif x.dtype == torch.float32 and slot.shape[-1] == 3 * _round_up(x.shape[-1], 8):
        ops.split3(x.contiguous(), 0, Cp=slot.shape[-1] // 3, out=slot)
    elif x.dtype == torch.float32 and x.shape[-1] == slot.shape[-1]:
        ops.cast_bf16(x.contiguous(), out=slot)
    else:
        slot.copy_(x)


def _conv_on(st: _Stream, key, module, x, **run_kw):
    """Feed tensor x ([T,H,W,C], bf16 or the fp32 trunk) to a conv that has no fused producer."""
    cs = st.conv(key, module)
    T, H, W, _ = x.shape
    _fill(cs.slot(T, H, W, x.device), x)
    return cs.run(**run_kw)


# A 96-channel convolution writes the NEXT layer's RMS norm + SiLU of its output itself (omh_conv_args.norm_*: in the
# stream kernel's epilogue, where one wave holds all 96 channels of a voxel) — the norm inside a block from conv1, the
# first norm of the next block (or of the head) from conv2, straight into that layer's convolution input slot.  The
# stand-alone kernel computes the same bits, so this is speed only (OMH_VAE_FUSE_NORM=0: always stand-alone).
_FUSE_NORM = os.environ.get("OMH_VAE_FUSE_NORM", "1") != "0"
# fp32-faithful mode: split-bf16 pairs on the stream kernel wherever it takes the layer (round 5); "0": three channel blocks
# everywhere (rounds 3-4; A/B timing, tests)
_PAIR = os.environ.get("OMH_VAE_PAIR", "1") != "0"
# ... with pairs a frame is 2 C channels, so the fp32 mode could take the bf16 mode's frame groups again (encoder chunks,
# decoder 2h x 2w stage: "e", "m" in OMH_VAE_F32_GROUPS).  Measured on one box (decode / encode frames/s): none 108.1 /
# 177.8, "e" 96.8 / 177.8, "m" 101.1 / 171.4: the larger groups do not pay in this mode — off by default
_F32_GROUPS = tuple(c in os.environ.get("OMH_VAE_F32_GROUPS", "") for c in "em")


def _res_block(st, key, blk: ResidualBlock, x, pre=False, nxt=None):
    """vae.py:202-220.  ``pre``: the producer of x already wrote norm1(x) into this block's first convolution slot.
    ``nxt`` = (gamma, conv state) of the norm + convolution that consume this block's output: returns (y, True) when
    that norm was written into the state's slot by this block's second convolution."""
    T, H, W, _ = x.shape
    dev = x.device
    h = x
    tf = st.trunk_f32
    fuse = _FUSE_NORM
    if not isinstance(blk.shortcut, nn.Identity):
        h = _conv_on(st, key + ".shortcut", blk.shortcut, x, out_f32=tf)
    ca = st.conv(key + ".residual.2", blk.residual[2])
    if not pre:
        st.norm(x, _gamma(blk.residual[0]), ca.slot(T, H, W, dev))
    cb = st.conv(key + ".residual.6", blk.residual[6])
    slot_b = cb.slot(T, H, W, dev)
    # fp32-faithful mode (round 5): the fused norm writes split-bf16 pairs, so both the producing convolution and the
    # consumer of the norm have to be pair layers of the stream kernel
    if fuse and ca.Cout == 96 and (not st.f32 or (ca.pair and cb.pair)):
        # inside the block: rounded once (bf16, or a bf16 pair), feeds one norm + conv
        ca.run(out_f32=st.f32, norm_gamma=_gamma(blk.residual[3]), norm_out=slot_b, norm_only=True)
    else:
        y = ca.run(out_f32=st.f32)
        st.norm(y, _gamma(blk.residual[3]), slot_b)
    if fuse and nxt is not None and cb.Cout == 96:
        gamma, cn = nxt
        slot_n = cn.slot(T, H, W, dev)
        if not st.f32 or (cb.pair and cn.pair):
            return cb.run(resid=h, out_f32=tf, norm_gamma=gamma, norm_out=slot_n), True
        y = cb.run(resid=h, out_f32=tf)
        st.norm(y, gamma, slot_n)                    # the slot is claimed: fill it here (the caller sees pre = True)
        return y, True
    return cb.run(resid=h, out_f32=tf), False


def _attention(st, key, blk: AttentionBlock, x):
    """vae.py:240-262 — per-frame single-head attention, D = C.  Scores go through the GEMM kernel
    (fp32 [HW, HW]) and a row-softmax kernel; it is 0.5 % of the decoder's work."""
    if st.f32:
        return _attention_f32(st, key, blk, x)
    T, H, W, Cc = x.shape
    HW = H * W
    HWp = _round_up(HW, 8)
    dev = x.device
    n = ops.rms_silu_cl(x, _gamma(blk.norm), do_silu=False).view(T * HW, Cc)
    wk = "attn:" + key
    if wk not in st.convs:
        wqkv = blk.to_qkv.weight.detach().float().view(3 * Cc, Cc)
        st.convs[wk] = (ops.cast_bf16(wqkv[:2 * Cc].contiguous()), blk.to_qkv.bias.detach().float()[:2 * Cc].contiguous(),
                        ops.cast_bf16(wqkv[2 * Cc:].contiguous()), blk.to_qkv.bias.detach().float()[2 * Cc:].contiguous())
    wqk, bqk, wv, bv = st.convs[wk]
    qk = ops.gemm(n, wqk, bias=bqk, epilogue=ops.EPI_BF16)                  # [T*HW, 2C]
    vt = torch.zeros(T, Cc, HWp, dtype=torch.bfloat16, device=dev)
    ops.gemm_raw(ops.ptr(wv), ops.ptr(n), ops.ptr(vt), Cc, HW, Cc, Cc, Cc, HWp, ops.EPI_BF16, bias=ops.ptr(bv),
                 bias_mode=ops.BIAS_M, batch=T, strideA=0, strideB=HW * Cc, strideC=Cc * HWp)
    s = torch.empty(T, HW, HW, dtype=torch.float32, device=dev)
    ops.gemm_raw(ops.ptr(qk), ops.ptr(qk, Cc), ops.ptr(s), HW, HW, Cc, 2 * Cc, 2 * Cc, HW, ops.EPI_F32, batch=T,
                 strideA=HW * 2 * Cc, strideB=HW * 2 * Cc, strideC=HW * HW)
    p = torch.zeros(T * HW, HWp, dtype=torch.bfloat16, device=dev)
    ops.softmax_rows(s.view(T * HW, HW), p, HW, 1.0 / math.sqrt(Cc))
    o = torch.empty(T, HW, Cc, dtype=torch.bfloat16, device=dev)
    ops.gemm_raw(ops.ptr(p), ops.ptr(vt), ops.ptr(o), HW, Cc, HWp, HWp, HWp, Cc, ops.EPI_BF16, batch=T,
                 strideA=HW * HWp, strideB=Cc * HWp, strideC=HW * Cc)
    del s, p
    cs = st.conv(key + ".proj", blk.proj)
    cs.slot(T, H, W, dev).copy_(o.view(T, H, W, Cc))
    return cs.run(resid=x, out_f32=st.trunk_f32)


def _attention_f32(st, key, blk: AttentionBlock, x):
    """The same block in the fp32-faithful mode: every GEMM operand a bf16 pair (activation-like operands in pattern 0,
    weight-like ones in pattern 1: q against k, P against V^T), fp32 results, an fp32 softmax."""
    T, H, W, Cc = x.shape
    HW = H * W
    HWp = _round_up(HW, 8)
    dev = x.device
    f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    n3 = ops.rms_silu_cl_split3(x, _gamma(blk.norm), do_silu=False).view(T * HW, 3 * Cc)       # [hi | lo | hi]
    wk = "attn3:" + key
    if wk not in st.convs:
        wqkv = blk.to_qkv.weight.detach().float().view(3 * Cc, Cc).contiguous()
        b = blk.to_qkv.bias.detach().float()
        st.convs[wk] = (ops.split3(wqkv[:2 * Cc], 1), b[:2 * Cc].contiguous(), ops.split3(wqkv[2 * Cc:], 1), b[2 * Cc:].contiguous())
    wqk, bqk, wv, bv = st.convs[wk]
    K3 = 3 * Cc
    qk = ops.gemm(n3, wqk, bias=bqk, epilogue=ops.EPI_F32)                                     # fp32 [T*HW, 2C]
    vt = torch.zeros(T, Cc, HWp, dtype=torch.float32, device=dev)                               # V^T = Wv n^T + bv
    ops.gemm_raw(ops.ptr(wv), ops.ptr(n3), ops.ptr(vt), Cc, HW, K3, K3, K3, HWp, ops.EPI_F32, bias=ops.ptr(bv),
                 bias_mode=ops.BIAS_M, batch=T, strideA=0, strideB=HW * K3, strideC=Cc * HWp)
    q3 = ops.split3(qk[:, :Cc], 0)                                                              # [T*HW, 3C]
    k3 = ops.split3(qk[:, Cc:], 1)
    s = f32(T, HW, HW)
    ops.gemm_raw(ops.ptr(q3), ops.ptr(k3), ops.ptr(s), HW, HW, K3, K3, K3, HW, ops.EPI_F32, batch=T,
                 strideA=HW * K3, strideB=HW * K3, strideC=HW * HW)
    p = torch.zeros(T * HW, HWp, dtype=torch.float32, device=dev)
    ops.softmax_rows_f32(s.view(T * HW, HW), p, HW, 1.0 / math.sqrt(Cc))
    p3 = ops.split3(p, 0, Cp=HWp)                                                               # [T*HW, 3 HWp]
    v3 = ops.split3(vt.view(T * Cc, HWp), 1, Cp=HWp)                                            # [T*C, 3 HWp]
    o = f32(T, HW, Cc)
    ops.gemm_raw(ops.ptr(p3), ops.ptr(v3), ops.ptr(o), HW, Cc, 3 * HWp, 3 * HWp, 3 * HWp, Cc, ops.EPI_F32, batch=T,
                 strideA=HW * 3 * HWp, strideB=Cc * 3 * HWp, strideC=HW * Cc)
    del s, p, p3
    cs = st.conv(key + ".proj", blk.proj)
    _fill(cs.slot(T, H, W, dev), o.view(T, H, W, Cc))
    return cs.run(resid=x, out_f32=True)


def _resample(st, key, rs: Resample, x):
    """vae.py:101-160."""
    T, H, W, Cc = x.shape
    if rs.mode in ("upsample2d", "upsample3d"):
        if rs.mode == "upsample3d":
            if key not in st.seen:
                st.seen.add(key)                    # first chunk: the reference's 'Rep' bypass (vae.py:106-108)
            else:
                x = _conv_on(st, key + ".time_conv", rs.time_conv, x, split_n=Cc, out_f32=st.f32)    # [2T, H, W, C]
        cs = st.conv(key + ".resample.1", rs.resample[1], up2=True)
        _fill(cs.slot(x.shape[0], H, W, x.device), x)
        return cs.run(out_f32=st.trunk_f32)
    if rs.mode in ("downsample2d", "downsample3d"):
        cs = st.conv(key + ".resample.1", rs.resample[1], stride_hw=2, pad=(0, 0))
        _fill(cs.slot(T, H, W, x.device), x)
        x = cs.run(out_f32=st.trunk_f32)
        if rs.mode == "downsample3d":
            tc = st.conv(key + ".time_conv", rs.time_conv)
            if key not in st.seen:
                st.seen.add(key)                    # first chunk passes through, remembered as history (vae.py:146-148)
                tc.slot(x.shape[0], x.shape[1], x.shape[2], x.device)
                tc.set_history(x[-1])
            else:
                _fill(tc.slot(x.shape[0], x.shape[1], x.shape[2], x.device), x)
                x = tc.run(out_f32=st.trunk_f32)
        return x
    return x


def _head(st, key, head: nn.Sequential, x, out_f32, pre=False):
    T, H, W, _ = x.shape
    cs = st.conv(key + ".2", head[2])
    if not pre:                                     # (pre: the last block's convolution wrote the norm into the slot)
        st.norm(x, _gamma(head[0]), cs.slot(T, H, W, x.device))
    return cs.run(out_f32=out_f32 or st.f32)


def _first_resample(seq):
    """Index of the first Resample in a layer list (len(seq) if none)."""
    for i, layer in enumerate(seq):
        if isinstance(layer, Resample):
            return i
    return len(seq)


def _last_resample(seq):
    """Index just past the last Resample in a layer list (0 if none)."""
    last = 0
    for i, layer in enumerate(seq):
        if isinstance(layer, Resample):
            last = i + 1
    return last


def _run_sequential(st, prefix, seq, x, start=0, stop=None, head=None):
    """Layers [start, stop) of ``seq``.  ``head`` = (key, nn.Sequential) of the head that follows the LAST layer of the
    range: with it the return value is (x, pre) — pre: the head's norm is already in its convolution slot (_head)."""
    last = (len(seq) if stop is None else min(stop, len(seq))) - 1
    pre = False
    for i, layer in enumerate(seq):
        if i < start or i > last:
            continue
        key = f"{prefix}.{i}"
        if isinstance(layer, ResidualBlock):
            nxt = None
            if i < last and isinstance(seq[i + 1], ResidualBlock):
                nb = seq[i + 1]
                nxt = (_gamma(nb.residual[0]), st.conv(f"{prefix}.{i + 1}.residual.2", nb.residual[2]))
            elif i == last and head is not None:
                nxt = (_gamma(head[1][0]), st.conv(head[0] + ".2", head[1][2]))
            x, pre = _res_block(st, key, layer, x, pre=pre, nxt=nxt)
        elif isinstance(layer, AttentionBlock):
            x, pre = _attention(st, key, layer, x), False
        elif isinstance(layer, Resample):
            x, pre = _resample(st, key, layer, x), False
        else:  # pragma: no cover
            raise TypeError(type(layer))
    return (x, pre) if head is not None else x


class WanVAE_(nn.Module):
    """vae.py:483-589."""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, dim_mult
        self.num_res_blocks, self.attn_scales = num_res_blocks, attn_scales
        self.temperal_downsample = temperal_downsample
        self.temperal_upsample = temperal_downsample[::-1]
        self.encoder = Encoder3d(dim, z_dim * 2, dim_mult, num_res_blocks, attn_scales, self.temperal_downsample,
                                 dropout)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(dim, z_dim, dim_mult, num_res_blocks, attn_scales, self.temperal_upsample, dropout)

    def _device(self):
        return self.conv1.weight.device

    def _f32(self):
        """True: the fp32-faithful arithmetic (split-bf16 operands) — set by the WanVAE wrapper from its ``dtype``
        (``compute_dtype`` attribute; a bare WanVAE_ computes with bf16 operands)."""
        return getattr(self, "compute_dtype", torch.bfloat16) == torch.float32

    @torch.no_grad()
    def encode(self, x, scale):
        """x fp32 [1, 3, T, H, W] -> mu [1, z, (T-1)/4+1, H/8, W/8] (vae.py:516-542)."""
        dev = self._device()
        if dev.type != "cuda":
            raise ops.OmhError("WanVAE runs on the MI355X only (no CPU fallback)")
        assert x.dim() == 5 and x.shape[0] == 1
        vid = x[0].to(device=dev, dtype=torch.float32).contiguous()
        _, T, H, W = vid.shape
        n_chunks = 1 + (T - 1) // 4
        st = _Stream(dev, f32=self._f32())
        enc = self.encoder
        out = None
        t_lat = 0
        # The chunked schedule (1, 4, 4, ... frames; vae.py:523-535) is kept where it matters — through the
        # Resample layers, whose first chunk is special.  Everything after the last Resample works at latent
        # resolution on ONE frame per chunk (6 240 voxels at 480x832: a quarter of the chip per convolution);
        # causal convolutions over [history | frames] give the same values whether the frames arrive one per call
        # or all together, so that tail runs once over all n_chunks frames.
        n_tail = _last_resample(enc.downsamples)
        rows = []
        i = 0
        while i < n_chunks:
            # the first frame alone (its chunk bypasses the temporal downsamples, vae.py:146-148), then _GROUP2 chunks
            # of 4 frames per step (causal convolutions over [history | frames]: same values as chunk by chunk)
            # (fp32 mode: one chunk — a [history | 8 frames] region of 3 x 96 channels at 480x832 would pass the 2 GiB
            # of 32-bit buffer offsets)
            # (... with the pair layout of round 5 a full-resolution frame is 2 x 96 channels = 153 MB: [history | 8 frames]
            # fits again, and so does the stride-2 Resample's three-block input of 8 frames)
            g = 1 if i == 0 else min(1 if (st.f32 and not (_PAIR and _F32_GROUPS[0])) else _GROUP2, n_chunks - i)
            t0, tn = (0, 1) if i == 0 else (1 + 4 * (i - 1), 4 * g)
            c1 = st.conv("encoder.conv1", enc.conv1)
            if st.f32:
                _fill(c1.slot(tn, H, W, dev), ops.nchw_to_cl_f32(vid, tn, t0, c1.Cin // 3))
            else:
                ops.nchw_to_cl(vid, tn, t0, c1.Cin, out=c1.slot(tn, H, W, dev))
            h = c1.run(out_f32=st.trunk_f32)
            rows.append(_run_sequential(st, "encoder.downsamples", enc.downsamples, h, stop=n_tail))
            i += g
        h = torch.cat(rows, dim=0) if len(rows) > 1 else rows[0]                   # [n_chunks, h, w, C]
        h = _run_sequential(st, "encoder.downsamples", enc.downsamples, h, start=n_tail)
        h = _run_sequential(st, "encoder.middle", enc.middle, h)
        h = _head(st, "encoder.head", enc.head, h, out_f32=False)                  # [t, h, w, 2z] (fp32 in the fp32 mode)
        mu = _conv_on(st, "conv1", self.conv1, h, out_f32=True)
        out = torch.empty(self.z_dim, n_chunks, h.shape[1], h.shape[2], dtype=torch.float32, device=dev)
        if isinstance(scale[0], torch.Tensor):
            add = (-scale[0]).to(device=dev, dtype=torch.float32).contiguous()
            mul = scale[1].to(device=dev, dtype=torch.float32).contiguous()
        else:
            add = torch.full((self.z_dim,), -float(scale[0]), device=dev)
            mul = torch.full((self.z_dim,), float(scale[1]), device=dev)
        ops.cl_to_nchw(mu, out, 0, self.z_dim, mul=mul, add=add)
        t_lat = mu.shape[0]
        return out[:, :t_lat].unsqueeze(0)

    @torch.no_grad()
    def decode(self, z, scale, clamp=None):
        """z [1, z, T', h, w] -> video fp32 [1, 3, 4(T'-1)+1, 8h, 8w] (vae.py:544-568)."""
        dev = self._device()
        if dev.type != "cuda":
            raise ops.OmhError("WanVAE runs on the MI355X only (no CPU fallback)")
        assert z.dim() == 5 and z.shape[0] == 1
        lat = z[0].to(device=dev, dtype=torch.float32).contiguous()
        _, Tl, h, w = lat.shape
        if isinstance(scale[0], torch.Tensor):
            mul = (1.0 / scale[1]).to(device=dev, dtype=torch.float32).contiguous()
            add = scale[0].to(device=dev, dtype=torch.float32).contiguous()
        else:
            mul = torch.full((self.z_dim,), 1.0 / float(scale[1]), device=dev)
            add = torch.full((self.z_dim,), float(scale[0]), device=dev)
        st = _Stream(dev, f32=self._f32())
        dec = self.decoder
        zc = _round_up(self.z_dim, 8)
        zcl = (ops.nchw_to_cl_f32 if st.f32 else ops.nchw_to_cl)(lat, Tl, 0, zc, mul=mul, add=add)   # z/scale1 + scale0
        x_all = _conv_on(st, "conv2", self.conv2, zcl, out_f32=st.f32)                     # [T', h, w, z]
        T_out = 4 * (Tl - 1) + 1
        out = torch.empty(3, T_out, 8 * h, 8 * w, dtype=torch.float32, device=dev)
        lo, hi = (-3.0e38, 3.0e38) if clamp is None else clamp
        t_pix = 0
        # Latent-resolution front (conv1, middle, the residual blocks before the first Resample) over ALL latent
        # frames in one pass — same values as frame by frame (causal convolutions, per-frame attention), 21x the
        # rows per launch; from the first Resample on, one latent frame per step as the reference (its first chunk
        # skips the temporal upsample, vae.py:105-125).
        n_front = _first_resample(dec.upsamples)
        c1 = st.conv("decoder.conv1", dec.conv1)
        _fill(c1.slot(Tl, h, w, dev), x_all)
        y_all = c1.run(out_f32=st.trunk_f32)
        y_all = _run_sequential(st, "decoder.middle", dec.middle, y_all)
        y_all = _run_sequential(st, "decoder.upsamples", dec.upsamples, y_all, stop=n_front)
        # From the first Resample to just past the second one (the 2h x 2w stage: 49 920 voxels per latent frame at
        # 480x832 — 195 workgroups of 256 rows on 256 CUs) the latent frames go _GROUP at a time, the first one
        # alone (its chunk skips the temporal upsamples); the full-resolution rest takes _GROUP2 latent frames per step
        # (the reference: one).  Causal convolutions over [history | frames]: same values either way.
        res_idx = [i for i, layer in enumerate(dec.upsamples) if isinstance(layer, Resample)]
        n_mid = res_idx[1] + 1 if len(res_idx) > 1 else len(dec.upsamples)
        i = 0
        while i < Tl:
            g = 1 if i == 0 else min(2 if (st.f32 and not (_PAIR and _F32_GROUPS[1])) else _GROUP, Tl - i)
            ymid = _run_sequential(st, "decoder.upsamples", dec.upsamples, y_all[i:i + g], start=n_front, stop=n_mid)
            per = ymid.shape[0] // g
            j = 0
            while j < g:                                      # the full-resolution rest: _GROUP2 latent frames per step
                g2 = min(1 if st.f32 else _GROUP2, g - j)     # (fp32 mode: the head's three-block input of 8 frames would pass 2 GiB)
                y, pre = _run_sequential(st, "decoder.upsamples", dec.upsamples, ymid[j * per:(j + g2) * per], start=n_mid,
                                         head=("decoder.head", dec.head))
                y = _head(st, "decoder.head", dec.head, y, out_f32=True, pre=pre)        # fp32 [t, 8h, 8w, 3]
                ops.cl_to_nchw(y, out, t_pix, 3, lo=lo, hi=hi)
                t_pix += y.shape[0]
                j += g2
            i += g
        assert t_pix == T_out
        return out.unsqueeze(0)

    def clear_cache(self):
        """The reference resets its per-call feature caches here (vae.py:582-589); this build keeps
        them in a per-call stream object, so there is nothing to clear.  Kept for API parity."""
        self._conv_num = count_conv3d(self.decoder)
        self._enc_conv_num = count_conv3d(self.encoder)


def _video_vae(pretrained_path=None, z_dim=None, device="cpu", **kwargs):
    """vae.py:592-616.  ``pretrained_path=None`` keeps the random initialisation (synthetic benchmarks)."""
    cfg = dict(dim=96, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
               temperal_downsample=[False, True, True], dropout=0.0)
    cfg.update(**kwargs)
    if pretrained_path is None:
        return WanVAE_(**cfg)
    with torch.device("meta"):
        model = WanVAE_(**cfg)
    logging.info(f"loading {pretrained_path}")
    model.load_state_dict(torch.load(pretrained_path, map_location=device), assign=True)
    return model


class WanVAE:
    """vae.py:619-663."""

    def __init__(self, z_dim=16, vae_pth="cache/vae_step_411000.pth", dtype=torch.float, device="cuda", **cfg):
        """``dtype`` selects the arithmetic class, as the reference's ``amp.autocast(dtype=self.dtype)`` does
        (vae.py:649-663): ``torch.float`` — the reference's default — computes every convolution / GEMM with split-bf16
        operand pairs and fp32 accumulation (an fp32-class product, three MFMA products per tile: <= 2e-4 from the fp32
        reference over the whole VAE); ``torch.bfloat16`` (what this package's pipelines and the benchmark pass) rounds
        the operands once to bf16 (1e-2 over the whole VAE, three times the throughput).  The latent mean / std stay
        fp32 in both."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"WanVAE dtype must be torch.float / torch.bfloat16 (torch.float16 = bfloat16 here), got {dtype}")
        self.dtype = dtype
        self.device = device
        mean = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
        std = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
        self.mean = torch.tensor(mean, dtype=torch.float32, device=device)
        self.std = torch.tensor(std, dtype=torch.float32, device=device)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = _video_vae(pretrained_path=vae_pth, z_dim=z_dim, **cfg).eval().requires_grad_(False).to(device)
        self.model.compute_dtype = torch.float32 if dtype == torch.float32 else torch.bfloat16

    def encode(self, videos):
        """videos: list of [3, T, H, W] -> list of fp32 [z, (T-1)/4+1, H/8, W/8]."""
        return [self.model.encode(u.unsqueeze(0), self.scale).float().squeeze(0) for u in videos]

    def decode(self, zs):
        """zs: list of [z, T', h, w] -> list of fp32 [3, 4(T'-1)+1, 8h, 8w] clamped to [-1, 1]."""
        return [self.model.decode(u.unsqueeze(0), self.scale, clamp=(-1.0, 1.0)).float().squeeze(0) for u in zs]


def bench_decode(latent, device, iters=1, telemetry=None, dtype=torch.bfloat16):
    """bench.py hook: frames/s of decoding one [16, T', 60, 104] latent, and of encoding the decoded clip back,
    with a random-init VAE.  Conv flops per frame from SURVEY.md section 8(d) (linear in H*W).  ``telemetry``: an
    object with start() / stop() -> dict (bench.Telemetry: shader clock and board power of the timed decodes)."""
    import time
    vae = WanVAE(vae_pth=None, device=device, dtype=dtype)
    z = latent.detach().float()
    z = (z - z.mean()) / z.std().clamp_min(1e-6)
    vae.decode([z[:, :2]])                     # warm-up: two chunks (first + steady state)
    torch.cuda.synchronize()
    if telemetry is not None:
        telemetry.start()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = vae.decode([z])[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    tele = telemetry.stop() if telemetry is not None else None
    frames = out.shape[1]
    area = (z.shape[2] * z.shape[3]) / (60 * 104)
    flops = (4.29 + (z.shape[1] - 1) * 13.49) * 1e12 * area
    res = {"frames_per_s": round(frames / dt, 2), "decode_s": round(dt, 3), "frames": int(frames), "repeats": iters,
           "conv_tflops": round(flops / dt / 1e12, 1), "mfma_roofline_frac": round(flops / dt / 2.5e15, 4),
           "finite": bool(torch.isfinite(out).all()), "weights": "random-init",
           "operands": "bf16 (one rounding per convolution operand), fp32 accumulate, fp32 residual trunk"
                       if dtype != torch.float32 else
                       "split-bf16 pairs (hi + lo, three MFMA products per tile), fp32 accumulate: WanVAE(dtype=torch.float)",
           "executed_mfma_tflops": round(flops * (3 if dtype == torch.float32 else 1) / dt / 1e12, 1)}
    if tele:
        res["decode_telemetry"] = tele
        res["mfma_frac_of_peak_at_measured_clock"] = round(flops / dt / 1e12 / tele["mfma_peak_at_mean_clock_tflops"], 4)
    vae.encode([out[:, :5]])                   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        lat = vae.encode([out])[0]
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / iters
    eflops = (2.66 + (z.shape[1] - 1) * 7.99) * 1e12 * area
    res.update({"encode_frames_per_s": round(frames / de, 2), "encode_s": round(de, 3),
                "encode_conv_tflops": round(eflops / de / 1e12, 1),
                "encode_finite": bool(torch.isfinite(lat).all()) and tuple(lat.shape) == tuple(z.shape)})
    return res
