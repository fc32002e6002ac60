"""CPU oracle (TEST INFRASTRUCTURE — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import
this) of the OmniHuman conditioning path of ``Omnihuman/omnihuman_wan_t2v.py`` (BASELINE config 4; SURVEY.md
section 8(f) rank 3): the audio / pose adapters, the reference-latent concatenation, CFG annealing and the
DPM-Solver++ sampling loop around the Wan DiT.  fp32 torch, function by function with reference file:line.

Pinning (see oracle/make_golden.py, tests/golden/omnihuman_adapters.npz): ``process_audio`` and the pose
Conv3d stack are checked against the reference's own ``OmniConditionsModule`` run in place.

What the reference cannot run, and what this oracle therefore DEFINES (product and oracle follow the same text):

* ``process_pose`` (:60-64 / :205-224) applies ``rearrange(x, 'b t c h w -> b t (c h w)')`` to the Conv3d output
  ``[B, C', T, h, w]`` — the labels are swapped, and ``pose_fc`` then fails with a shape error for every T != C'.
  Definition: heat-maps enter as ``[B, K, T, H, W]`` (the only layout ``nn.Conv3d(K, ...)`` accepts), the conv output
  is permuted to ``[B, T, C', h, w]`` and flattened per frame — one pose token per frame, as the comments say.
* ``WanModel.forward`` has no ``extra_conditions`` argument (model.py:502), so the conditional call (:408-414)
  raises ``TypeError``; ``condition_projector`` and ``temporal_embed`` are never applied.  Definition
  (``condition_tokens``): every modality token gets the temporal embedding of its frame, goes through
  ``condition_projector``, and the resulting ``[B, Ne, dim]`` tokens are prepended to the embedded text context, so
  each block's cross-attention attends to them (the mechanism the i2v backbone uses for its CLIP tokens,
  model.py:534-537).  Audio tokens are ``2*dim`` wide after the adjacent-frame concat (:197-201): the two halves are
  the tokens of frame t and t+1 and are projected as such.
* the reference latent is concatenated on dim 2 of a batched 5-D latent although ``vae.encode`` returns a 4-D one
  (:390-392), ``seq_len`` ignores the extra frame (:399) and the prediction is not cropped back before
  ``scheduler.step`` (:432).  Definition: per sample ``cat([ref [16,1,h,w], latent [16,T,h,w]], dim=1)``,
  ``seq_len`` of that, and the last T frames of the DiT output are the velocity.
"""
import torch
import torch.nn.functional as F

from . import wan_dit_oracle as O
from .sampler_oracle import DPMSolverOracle


def process_audio(sd, audio, prefix="audio_processor."):
    """omnihuman_wan_t2v.py:53-58 / :189-203 — Linear, SiLU, Linear; then [tok[:-1] | tok[1:]] on the channel axis."""
    h = F.linear(audio.float(), sd[prefix + "0.weight"], sd[prefix + "0.bias"])
    tok = F.linear(F.silu(h), sd[prefix + "2.weight"], sd[prefix + "2.bias"])
    if tok.shape[1] > 1:
        tok = torch.cat([tok[:, :-1], tok[:, 1:]], dim=-1)
    return tok


def pose_conv_stack(sd, pose, prefix="pose_processor.", gates=None):
    """:37-45 / :149-157 — three Conv3d(3x3x3, padding 1) + ReLU, spatial stride 1, 2, 2.  pose [B, K, T, H, W].
    ``gates`` (tests only): three boolean masks of the layers' output shapes that REPLACE the ReLUs' own on/off
    decisions (y = conv * gate) — the ReLU is discontinuous in its gradient at 0, and a test that wants to compare
    gradients with a lower-precision forward hands over that forward's decisions."""
    def act(x, i):
        return F.relu(x) if gates is None else x * gates[i].to(x.dtype)
    x = act(F.conv3d(pose.float(), sd[prefix + "0.weight"], sd[prefix + "0.bias"], padding=1), 0)
    x = act(F.conv3d(x, sd[prefix + "2.weight"], sd[prefix + "2.bias"], stride=(1, 2, 2), padding=1), 1)
    return act(F.conv3d(x, sd[prefix + "4.weight"], sd[prefix + "4.bias"], stride=(1, 2, 2), padding=1), 2)


def process_pose(sd, pose, prefix="pose_processor.", gates=None):
    """:205-224 with the frame / channel axes in the order the text describes (module docstring)."""
    x = pose_conv_stack(sd, pose, prefix, gates)                 # [B, C', T, h, w]
    x = x.permute(0, 2, 1, 3, 4).flatten(2)                      # [B, T, C'*h*w]
    return F.linear(x, sd["pose_fc.weight"], sd["pose_fc.bias"])


def condition_tokens(sd, audio_tokens=None, pose_tokens=None):
    """[B, Ne, dim] cross-attention tokens from the adapter outputs (module docstring): audio pairs first (frame t
    then t+1 for every pair), then pose, each ``condition_projector(token + temporal_embed[frame])``."""
    te = sd["temporal_embed"][0]                                  # [num_frames, dim]
    d = te.shape[1]
    toks = []
    if audio_tokens is not None:
        B, Tm1, w = audio_tokens.shape
        if w == 2 * d:
            a = audio_tokens.view(B, Tm1, 2, d) + torch.stack([te[:Tm1], te[1:Tm1 + 1]], dim=1)
            toks.append(a.reshape(B, 2 * Tm1, d))
        else:                                                     # a single frame: no adjacent-frame concat (:196)
            toks.append(audio_tokens + te[:Tm1])
    if pose_tokens is not None:
        toks.append(pose_tokens + te[:pose_tokens.shape[1]])
    if not toks:
        return None
    t = torch.cat(toks, dim=1)
    return F.linear(t, sd["condition_projector.weight"], sd["condition_projector.bias"])


def annealed_cfg(i, n_steps, cfg_scale):
    """:425-428 — linear from cfg_scale at step 0 towards 1."""
    progress = i / n_steps
    return cfg_scale * (1.0 - progress) + 1.0 * progress


@torch.no_grad()
def sample(dit_sd, dit_cfg, omni_sd, noise, text_ctx, text_ctx_null, reference_latent=None, audio=None, pose=None,
           num_inference_steps=4, cfg_scale=7.5, shift=1.0):
    """:364-433 — the multi-step loop: unconditional forward on the null text (no extra tokens), conditional forward
    with text + condition tokens, annealed CFG, FlowDPMSolverMultistepScheduler (solver_order 2, shift 1.0, :173-178)
    step.  ``noise`` [16, T, h, w]; returns the final latent."""
    a = process_audio(omni_sd, audio) if audio is not None else None
    p = process_pose(omni_sd, pose) if pose is not None else None
    extra = condition_tokens(omni_sd, a, p)
    sch = DPMSolverOracle(num_inference_steps, shift, default_schedule=True)
    lat = noise.clone().float()
    T = lat.shape[1]
    for i, t in enumerate(sch.timesteps):
        x = lat if reference_latent is None else torch.cat([reference_latent.float(), lat], dim=1)
        pt, ph, pw = dit_cfg.patch_size
        seq_len = (x.shape[1] // pt) * (x.shape[2] // ph) * (x.shape[3] // pw)
        tt = torch.tensor([float(t)])
        u = O.dit_forward(dit_sd, dit_cfg, [x], tt, [text_ctx_null], seq_len)[0][:, -T:]
        c = O.dit_forward(dit_sd, dit_cfg, [x], tt, [text_ctx], seq_len, extra_tokens=extra)[0][:, -T:]
        g = annealed_cfg(i, num_inference_steps, cfg_scale)
        lat = sch.step(u + g * (c - u), lat)
    return lat
