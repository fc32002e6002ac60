"""``OmniConditionsModule`` / ``OmniHumanWanT2V`` — the conditioning path of
Omnihuman/omnihuman_wan_t2v.py (BASELINE config 4) on the gfx950 kernels.

Same module tree and parameter names as the reference (audio_processor, pose_guider / pose_processor, pose_fc,
temporal_embed, condition_projector), so its state dicts load; the arithmetic runs on libomh.so:

* audio MLP and the condition projector: ``omh_dense_f32`` (fp32, SiLU fused on the second layer's input);
* pose guider: the three Conv3d(3x3x3, padding 1, spatial stride 1/2/2) layers run on the VAE's implicit-GEMM
  convolution ``omh_conv_cl_bf16`` (channels-last bf16, MFMA).  That kernel is causal in time — frame t reads
  buffer frames t..t+2 — so symmetric temporal padding is one zero frame in front and one behind the clip;
  ReLU in place (``omh_relu_bf16``); ``pose_fc`` on the bf16 GEMM;
* the denoising loop: the DiT forward (WanModel), the fused CFG + DPM-Solver++ kernel with the annealed
  guidance scale, the VAE for the reference image and the final decode.

The reference file cannot run end to end (see oracle/omnihuman_oracle.py, which lists what fails and the
definition used instead); this module implements exactly that oracle's text: pose heat-maps are
``[B, K, T, H, W]``, one pose token per frame; condition tokens = ``condition_projector(token +
temporal_embed[frame])`` prepended to the text context of the conditional forward
(``WanModel.forward(extra_conditions=...)``); the reference latent is one extra leading latent frame whose
prediction is dropped.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler

__all__ = ["OmniConditionsModule", "OmniHumanWanT2V"]


def _ru(a, b):
    return (a + b - 1) // b * b


# ----------------------------------------------------------------------------- kernels behind the adapters
# Each adapter layer is an autograd node whose forward AND backward run on libomh.so (the training step of
# omnihuman_wan_t2v.py:453-488 back-propagates the flow-matching loss through the DiT's cross-attention into the
# condition tokens and from there into these layers); under torch.no_grad() they are plain kernel launches.
class _DenseFn(torch.autograd.Function):
    """y = act_in(x) W^T + b, fp32 (omh_dense_f32 / omh_dense_f32_bwd); act_in: 0 none, 1 SiLU."""

    @staticmethod
    def forward(ctx, x, w, b, act_in):
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        wf = w.detach().float().contiguous()
        y = ops.dense_f32(x2, wf, b.detach().float().contiguous(), act_in, 0)
        ctx.save_for_backward(x2, wf)
        ctx.act_in, ctx.shape, ctx.wdt = act_in, x.shape, w.dtype
        return y.view(*x.shape[:-1], -1)

    @staticmethod
    def backward(ctx, dy):
        x2, wf = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).float().contiguous()
        dW, db = torch.zeros_like(wf), torch.zeros(wf.shape[0], dtype=torch.float32, device=wf.device)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        ops.dense_f32_bwd(x2, wf, dy2, dW=dW, db=db, dx=dx, act_in=ctx.act_in)
        return (None if dx is None else dx.view(ctx.shape)), dW.to(ctx.wdt), db.to(ctx.wdt), None


def _mlp_silu(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Linear -> SiLU -> Linear on [..., K] fp32 (omnihuman_wan_t2v.py:30-34)."""
    l0, l2 = seq[0], seq[2]
    return _DenseFn.apply(_DenseFn.apply(x, l0.weight, l0.bias, 0), l2.weight, l2.bias, 1)


def _pack_conv(conv: nn.Conv3d, cin_p: int) -> torch.Tensor:
    """[Cout, Cin, kt, kh, kw] fp32 -> bf16 [Cout, kt*kh*kw*cin_p] (tap-major, channel-minor, Cin zero-padded)."""
    w = conv.weight.detach().float()
    co, ci = w.shape[:2]
    wp = torch.zeros(co, 3, 3, 3, cin_p, dtype=torch.float32, device=w.device)
    wp[..., :ci] = w.permute(0, 2, 3, 4, 1)
    return ops.cast_bf16(wp.reshape(co, -1).contiguous())


def _conv3d_relu_fwd(x_cl: torch.Tensor, conv: nn.Conv3d, stride_hw: int) -> torch.Tensor:
    """x_cl bf16 [T, H, W, Cp] -> ReLU(Conv3d(k=3, padding=1, stride (1, s, s))) as bf16 [T, H', W', Cout_p].
    Output channels are padded to a multiple of 8 (zero weights, zero bias) so the result feeds the next layer."""
    T, H, W, Cp = x_cl.shape
    co = conv.out_channels
    co_p = _ru(co, 8)
    w = _pack_conv(conv, Cp)
    b = conv.bias.detach().float().contiguous()
    if co_p != co:
        w = torch.cat([w, torch.zeros(co_p - co, w.shape[1], dtype=w.dtype, device=w.device)]).contiguous()
        b = torch.cat([b, torch.zeros(co_p - co, device=b.device)]).contiguous()
    buf = torch.zeros(T + 2, H, W, Cp, dtype=torch.bfloat16, device=x_cl.device)     # [0, x_0 .. x_{T-1}, 0]
    buf[1:T + 1].copy_(x_cl)
    Ho, Wo = (H + 2 - 3) // stride_hw + 1, (W + 2 - 3) // stride_hw + 1
    y = ops.conv_cl(buf, w, b, T, Ho, Wo, co_p, 3, 3, 3, stride_t=1, stride_hw=stride_hw, pad_h=1, pad_w=1)
    return ops.relu_bf16_(y)


def _conv3d_relu_bwd(x_cl, y, dy, conv: nn.Conv3d, s: int, need_dx: bool):
    """Backward of ``_conv3d_relu_fwd``: (dx_cl or None, dW [Cout, Cin, 3, 3, 3] fp32, db [Cout] fp32).

    With g = dy where y > 0 (``omh_relu_bwd_bf16``) scattered onto the INPUT grid at (t, s*ho, s*wo) (zeros elsewhere):
      * bias:   column sums of g (``omh_colsum_accum``);
      * weight: dW[co, tap, ci] = sum_voxels g[v, co] x_pad[v + offset(tap), ci] — with g and the zero-padded input on
        the same (H+2, W+2) row pitch a tap is a constant row offset, so each of the 27 taps is one ``omh_gemm_bf16_tn``
        over the voxels (the contraction index on the rows of both operands, no transposes);
      * input:  the transposed convolution = a stride-1 "same" convolution of the scattered g with the kernel
        flipped along t, h, w and in/out channels swapped — the forward's own implicit-GEMM kernel."""
    T, H, W, Cp = x_cl.shape
    _, Ho, Wo, Cop = y.shape
    co, ci = conv.out_channels, conv.in_channels
    dev = x_cl.device
    g = ops.relu_bwd_bf16(dy.contiguous(), y)
    db = torch.zeros(Cop, dtype=torch.float32, device=dev)
    ops.colsum_accum(g.view(-1, Cop), db)
    P = (H + 2) * (W + 2)
    G = torch.zeros(T, H + 2, W + 2, Cop, dtype=torch.bfloat16, device=dev)
    G[:, 0:s * Ho:s, 0:s * Wo:s] = g                                     # scatter (data movement only)
    tail = 2 * (W + 2) + 8
    xflat = torch.zeros((T + 2) * P + tail, Cp, dtype=torch.bfloat16, device=dev)
    xflat[:(T + 2) * P].view(T + 2, H + 2, W + 2, Cp)[1:T + 1, 1:H + 1, 1:W + 1] = x_cl
    dWp = torch.empty(Cop, 27, Cp, dtype=torch.float32, device=dev)
    Gf = G.view(T * P, Cop)
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                off = kt * P + kh * (W + 2) + kw
                ops.gemm_tn(Gf, xflat[off:off + T * P], out=dWp[:, (kt * 3 + kh) * 3 + kw, :])
    dW = dWp[:co, :, :ci].reshape(co, 3, 3, 3, ci).permute(0, 4, 1, 2, 3).contiguous()
    dx = None
    if need_dx:
        w = conv.weight.detach().float()                                 # [co, ci, kt, kh, kw]
        wf = torch.zeros(Cp, 3, 3, 3, Cop, dtype=torch.float32, device=dev)
        wf[:ci, :, :, :, :co] = w.flip(2, 3, 4).permute(1, 2, 3, 4, 0)
        buf = torch.zeros(T + 2, H, W, Cop, dtype=torch.bfloat16, device=dev)
        buf[1:T + 1] = G[:, :H, :W]
        dx = ops.conv_cl(buf, ops.cast_bf16(wf.reshape(Cp, -1).contiguous()), None, T, H, W, Cp, 3, 3, 3, stride_t=1,
                         stride_hw=1, pad_h=1, pad_w=1)
    return dx, dW, db[:co].contiguous()


class _Conv3dReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, weight, bias, conv, stride_hw):
        y = _conv3d_relu_fwd(x_cl, conv, stride_hw)
        ctx.save_for_backward(x_cl, y)
        ctx.conv, ctx.s = conv, stride_hw
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, y = ctx.saved_tensors
        dx, dW, db = _conv3d_relu_bwd(x_cl, y, dy, ctx.conv, ctx.s, ctx.needs_input_grad[0])
        return dx, dW.to(ctx.conv.weight.dtype), db.to(ctx.conv.bias.dtype), None, None


def _conv3d_relu(x_cl: torch.Tensor, conv: nn.Conv3d, stride_hw: int) -> torch.Tensor:
    return _Conv3dReluFn.apply(x_cl, conv.weight, conv.bias, conv, stride_hw)


class _LinearBf16Fn(torch.autograd.Function):
    """y = x W^T + b on the bf16 MFMA GEMM (fp32 result); backward: dx = dy W (k-major B), dW = dy^T x (TN GEMM)."""

    @staticmethod
    def forward(ctx, x, w, b):
        a = ops.cast_bf16(x.float().contiguous())
        wb = ops.cast_bf16(w.detach().float().contiguous())
        ctx.save_for_backward(a, wb)
        ctx.wdt = w.dtype
        return ops.gemm(a, wb, bias=b.detach().float().contiguous(), epilogue=ops.EPI_F32)

    @staticmethod
    def backward(ctx, dy):
        a, wb = ctx.saved_tensors
        dyf = dy.float().contiguous()
        dyb = ops.cast_bf16(dyf)
        db = torch.zeros(wb.shape[0], dtype=torch.float32, device=dy.device)
        ops.colsum_accum(dyf, db)
        dW = ops.gemm_tn(dyb, a)                                          # [N, K] = dy^T x
        dx = ops.gemm(dyb, wb, epilogue=ops.EPI_F32, b_kmajor=True) if ctx.needs_input_grad[0] else None
        return dx, dW.to(ctx.wdt), db.to(ctx.wdt)


def _pose_stack(convs, pose: torch.Tensor) -> torch.Tensor:
    """pose fp32 [K, T, H, W] -> fp32 [T, C', h, w] through the three conv+ReLU layers."""
    K, T, H, W = pose.shape
    x = ops.nchw_to_cl(pose.float().contiguous(), T, 0, _ru(K, 8))
    for conv, s in zip(convs, (1, 2, 2)):
        x = _conv3d_relu(x, conv, s)
    return x[..., :convs[-1].out_channels].permute(0, 3, 1, 2).float()


def _linear_bf16(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """x fp32 [R, K] -> fp32 [R, N] on the bf16 MFMA GEMM (K % 8 == 0)."""
    return _LinearBf16Fn.apply(x, lin.weight, lin.bias)


class _Adapters:
    """The adapter arithmetic shared by both classes; ``self._pose_convs()`` names the Conv3d layers."""

    def process_audio(self, audio_features: torch.Tensor) -> torch.Tensor:
        """[B, T, audio_dim] -> [B, T-1, 2*model_dim]  (omnihuman_wan_t2v.py:53-58 / :189-203)."""
        tok = _mlp_silu(self.audio_processor, audio_features)
        if tok.shape[1] > 1:
            tok = torch.cat([tok[:, :-1], tok[:, 1:]], dim=-1)
        return tok

    def pose_features(self, pose_heatmaps: torch.Tensor) -> torch.Tensor:
        """[B, K, T, H, W] -> [B, C', T, H/4, W/4]: the Conv3d stack alone, in the reference's output layout."""
        convs = self._pose_convs()
        return torch.stack([_pose_stack(convs, p).permute(1, 0, 2, 3) for p in pose_heatmaps])

    def process_pose(self, pose_heatmaps: torch.Tensor) -> torch.Tensor:
        """[B, K, T, H, W] -> [B, T, model_dim], one token per frame (:60-64 / :205-224 as the oracle defines it)."""
        convs = self._pose_convs()
        out = []
        for p in pose_heatmaps:
            f = _pose_stack(convs, p)                                   # [T, C', h, w]
            out.append(_linear_bf16(self.pose_fc, f.flatten(1)))
        return torch.stack(out)

    def condition_tokens(self, audio_tokens: Optional[torch.Tensor] = None,
                         pose_tokens: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """[B, Ne, model_dim] cross-attention tokens: audio pairs (frame t, then t+1), then pose, each
        ``condition_projector(token + temporal_embed[frame])`` (oracle/omnihuman_oracle.py:condition_tokens)."""
        te = self.temporal_embed[0].float()          # (element-wise adds on O(100) tokens: host-side glue in torch)
        d = te.shape[1]
        toks = []
        if audio_tokens is not None:
            B, Tm1, w = audio_tokens.shape
            if w == 2 * d:
                a = audio_tokens.view(B, Tm1, 2, d) + torch.stack([te[:Tm1], te[1:Tm1 + 1]], dim=1)
                toks.append(a.reshape(B, 2 * Tm1, d))
            else:
                toks.append(audio_tokens + te[:Tm1])
        if pose_tokens is not None:
            toks.append(pose_tokens + te[:pose_tokens.shape[1]])
        if not toks:
            return None
        t = torch.cat(toks, dim=1).contiguous()
        cp = self.condition_projector
        return _DenseFn.apply(t, cp.weight, cp.bias, 0)


class OmniConditionsModule(_Adapters, nn.Module):
    """omnihuman_wan_t2v.py:13-92 (same constructor, parameters and methods)."""

    def __init__(self, model_dim: int = 5120, num_frames: int = 49, audio_dim: int = 1024, pose_keypoints: int = 33,
                 device=None, dtype=None):
        super().__init__()
        self.device, self.dtype = device, dtype
        kw = dict(device=device, dtype=dtype)
        self.audio_processor = nn.Sequential(nn.Linear(audio_dim, model_dim, **kw), nn.SiLU(),
                                             nn.Linear(model_dim, model_dim, **kw))
        self.pose_guider = nn.Sequential(
            nn.Conv3d(pose_keypoints, 64, kernel_size=(3, 3, 3), padding=1, **kw), nn.ReLU(),
            nn.Conv3d(64, 128, kernel_size=(3, 3, 3), stride=(1, 2, 2), padding=1, **kw), nn.ReLU(),
            nn.Conv3d(128, model_dim // 4, kernel_size=(3, 3, 3), stride=(1, 2, 2), padding=1, **kw), nn.ReLU())
        self.pose_fc = nn.Linear((model_dim // 4) * 16 * 16, model_dim, **kw)
        self.temporal_embed = nn.Parameter(torch.randn(1, num_frames, model_dim, **kw) / (model_dim ** 0.5))
        self.condition_projector = nn.Linear(model_dim, model_dim, **kw)

    def _pose_convs(self):
        return [self.pose_guider[0], self.pose_guider[2], self.pose_guider[4]]

    def process_reference(self, reference_image: torch.Tensor, vae) -> torch.Tensor:
        with torch.no_grad():
            return vae.encode([reference_image])[0]

    @torch.no_grad()
    def forward(self, audio=None, pose=None, text_embeddings=None, reference_latent=None) -> Dict[str, torch.Tensor]:
        cond = {}
        first = next((t for t in (audio, pose, text_embeddings, reference_latent) if t is not None), None)
        batch_size = first.shape[0] if first is not None else 1
        if audio is not None:
            cond["audio"] = self.process_audio(audio)
        if pose is not None:
            cond["pose"] = self.process_pose(pose)
        if text_embeddings is not None:
            cond["text"] = text_embeddings
        if reference_latent is not None:
            cond["reference"] = reference_latent
        cond["temporal"] = self.temporal_embed.detach().expand(batch_size, -1, -1)
        return cond


class OmniHumanWanT2V(_Adapters, nn.Module):
    """omnihuman_wan_t2v.py:94-470.  ``wan_t2v`` is a ``wan.text2video.WanT2V`` (or any object with ``.model``,
    ``.vae`` and, for string prompts, ``.text_encoder``); without it the reference's constructor arguments are
    used to build one (``config['wan_config']``, ``config['checkpoint_dir']``).  Text conditioning may be given as
    prompts (needs a text encoder) or directly as umT5 embeddings (``text_context`` / ``text_context_null``)."""

    def __init__(self, config: dict, device_id: int = 0, wan_t2v=None):
        super().__init__()
        self.config = config
        self.device = torch.device(f"cuda:{device_id}")
        self.num_frames = config.get("num_frames", 49)
        self.num_keypoints = config.get("num_keypoints", 308)
        if wan_t2v is None:
            from .wan import WanT2V
            from .wan.configs import t2v_14B
            wan_t2v = WanT2V(config=config.get("wan_config", t2v_14B),
                             checkpoint_dir=config.get("checkpoint_dir", "./checkpoints"), device_id=device_id, rank=0,
                             t5_fsdp=False, dit_fsdp=False, use_usp=False, t5_cpu=config.get("t5_cpu", False))
        self.wan_t2v = wan_t2v
        self.wan_t2v.model.to(self.device)
        self._init_condition_processors()
        self._init_diffusion_scheduler()

    def _init_condition_processors(self):
        """:136-170."""
        d = self.config.get("model_dim", 5120)
        dev = self.device
        self.audio_processor = nn.Sequential(nn.Linear(self.config.get("audio_dim", 1024), d, device=dev), nn.SiLU(),
                                             nn.Linear(d, d, device=dev))
        self.pose_processor = nn.Sequential(
            nn.Conv3d(self.num_keypoints, 128, kernel_size=(3, 3, 3), padding=1, device=dev), nn.ReLU(),
            nn.Conv3d(128, 256, kernel_size=(3, 3, 3), stride=(1, 2, 2), padding=1, device=dev), nn.ReLU(),
            nn.Conv3d(256, d // 4, kernel_size=(3, 3, 3), stride=(1, 2, 2), padding=1, device=dev), nn.ReLU())
        self.pose_fc = nn.Linear((d // 4) * 16 * 16, d, device=dev)
        self.temporal_embed = nn.Parameter(torch.randn(1, self.num_frames, d, device=dev) / (d ** 0.5))
        self.condition_projector = nn.Linear(d, d, device=dev)

    def _init_diffusion_scheduler(self):
        """:172-180."""
        self.scheduler = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=2,
                                                         prediction_type="flow_prediction", shift=1.0)

    def _pose_convs(self):
        return [self.pose_processor[0], self.pose_processor[2], self.pose_processor[4]]

    def process_reference(self, reference_image: torch.Tensor) -> torch.Tensor:
        """:226-239 — ``[3, 1, H, W]`` (or ``[3, H, W]``) image in [-1, 1] -> latent ``[16, 1, H/8, W/8]``."""
        img = reference_image if reference_image.dim() == 4 else reference_image[:, None]
        with torch.no_grad():
            return self.wan_t2v.vae.encode([img.to(self.device)])[0]

    def _text(self, prompt):
        enc = getattr(self.wan_t2v, "text_encoder", None)
        if enc is None:
            raise RuntimeError("no text encoder attached to wan_t2v: pass text_context= / text_context_null=")
        return enc([prompt], self.device)[0]

    @torch.no_grad()
    def prepare_conditions(self, text_prompt=None, audio=None, pose=None, reference_image=None,
                           text_context: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """:241-299."""
        cond = {}
        batch_size = 1
        if audio is not None:
            batch_size = audio.shape[0]
        elif pose is not None:
            batch_size = pose.shape[0]
        if text_context is not None:
            cond["text"] = text_context.to(self.device)
        elif text_prompt is not None:
            cond["text"] = self._text(text_prompt)
        if audio is not None:
            cond["audio"] = self.process_audio(audio.to(self.device))
        if pose is not None:
            cond["pose"] = self.process_pose(pose.to(self.device))
        if reference_image is not None:
            cond["reference"] = self.process_reference(reference_image)
        cond["temporal"] = self.temporal_embed.detach().expand(batch_size, -1, -1)
        tok = self.condition_tokens(cond.get("audio"), cond.get("pose"))
        if tok is not None:
            cond["tokens"] = tok
        return cond

    def _compute_seq_len(self, shape) -> int:
        """:301-311 — shape [B, C, T, H, W]."""
        p = self.wan_t2v.model.patch_size
        return (shape[2] // p[0]) * (shape[3] // p[1]) * (shape[4] // p[2])

    @torch.no_grad()
    def forward(self, text_prompt=None, audio=None, pose=None, reference_image=None, num_inference_steps: int = 50,
                cfg_scale: float = 7.5, seed: int = -1, text_context=None, text_context_null=None,
                latent_hw=None, noise: Optional[torch.Tensor] = None, return_latent: bool = False) -> torch.Tensor:
        """:313-438 — multi-step sampling with annealed CFG; returns the decoded video ``[3, N, H, W]``."""
        cond = self.prepare_conditions(text_prompt, audio, pose, reference_image, text_context)
        ctx = cond.get("text")
        if ctx is None:
            raise ValueError("text conditioning is required (the reference passes context=None to WanModel, which "
                             "cannot embed it, omnihuman_wan_t2v.py:408)")
        if text_context_null is None:
            text_context_null = self._text(self.config.get("negative_prompt", ""))
        ctx_null = text_context_null.to(self.device)
        ref = cond.get("reference")
        if ref is not None:
            h, w = ref.shape[-2:]
        elif latent_hw is not None:
            h, w = latent_hw
        else:
            h, w = self.config.get("latent_height", 64), self.config.get("latent_width", 64)
        model, vae = self.wan_t2v.model, self.wan_t2v.vae
        z = getattr(getattr(vae, "model", None), "z_dim", 16)
        T = self.num_frames
        if noise is None:
            if seed < 0:
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            g = torch.Generator(device=self.device).manual_seed(seed)
            noise = torch.randn(z, T, h, w, device=self.device, generator=g, dtype=torch.float32)
        lat = noise.to(self.device).float()
        T = lat.shape[1]
        self.scheduler.set_timesteps(num_inference_steps, device=self.device)
        self.scheduler.set_begin_index(0)
        seq_len = self._compute_seq_len((1, z, T + (0 if ref is None else ref.shape[1]), h, w))
        # everything that depends on the conditions alone: once per sample
        st_c = model.encode_context([ctx], extra_conditions=cond.get("tokens"))
        st_u = model.encode_context([ctx_null])
        n = len(self.scheduler.timesteps)
        for i, t in enumerate(self.scheduler.timesteps):
            x = lat if ref is None else torch.cat([ref.float(), lat], dim=1)
            ts = t.reshape(1).to(self.device)
            u = model([x], ts, st_u, seq_len)[0][:, -T:].contiguous()
            c = model([x], ts, st_c, seq_len)[0][:, -T:].contiguous()
            progress = i / n
            current_cfg = cfg_scale * (1.0 - progress) + 1.0 * progress                      # :425-428
            lat = self.scheduler.step_cfg(c, u, current_cfg, lat)
        return lat if return_latent else vae.decode([lat])[0]

    def training_step(self, frames: torch.Tensor, conditions: Dict[str, torch.Tensor], t: torch.Tensor,
                      noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """:453-488 — the flow-matching loss ``mean((pred - frames)^2 (1 - t))`` of the conditioned backbone on
        ``noisy = (1 - t) frames + t noise``, differentiable end to end: autograd runs through the hand-written
        backward of the DiT (wan/modules/model_train.py), out of its cross-attention into the condition tokens, and
        through ``condition_projector`` / ``temporal_embed`` into the audio MLP and the pose Conv3d stack + ``pose_fc``
        (every layer's backward on libomh.so, see ``_DenseFn`` / ``_Conv3dReluFn`` / ``_LinearBf16Fn``).

        ``conditions``: ``text`` [L, 4096] (required: WanModel cannot embed ``None``), and any of
        ``tokens`` [B, Ne, dim] (ready condition tokens, e.g. from ``prepare_conditions``),
        ``audio`` / ``pose`` (adapter outputs, turned into tokens here, gradients reach the projector and the
        temporal embedding), ``audio_features`` [B, T, audio_dim] / ``pose_heatmaps`` [B, K, T, H, W] (raw inputs:
        the adapters run here and are trained).  ``noise`` fixes the draw (tests)."""
        frames = frames.to(self.device).float()
        t = t.to(self.device).float()
        if noise is None:
            noise = torch.randn_like(frames)
        tt = t.view(-1, 1, 1, 1, 1)
        noisy = (1 - tt) * frames + tt * noise.to(self.device)
        ctx = conditions.get("text")
        if ctx is None:
            raise ValueError("text conditioning is required (omnihuman_wan_t2v.py:476-482 passes [None] otherwise, "
                             "which WanModel cannot embed)")
        audio, pose = conditions.get("audio"), conditions.get("pose")
        if conditions.get("audio_features") is not None:
            audio = self.process_audio(conditions["audio_features"].to(self.device))
        if conditions.get("pose_heatmaps") is not None:
            pose = self.process_pose(conditions["pose_heatmaps"].to(self.device))
        tokens = conditions.get("tokens")
        if tokens is None:
            tokens = self.condition_tokens(audio, pose)
        B = frames.shape[0]
        pred = self.wan_t2v.model(noisy, t, context=[ctx.to(self.device)] * B,
                                  seq_len=self._compute_seq_len(noisy.shape), extra_conditions=tokens)
        pred = torch.stack(pred)
        return torch.mean((pred - frames) ** 2 * (1 - tt))
