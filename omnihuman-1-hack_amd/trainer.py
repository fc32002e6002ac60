"""The student training step of seaweed_apt/distilled_trainer.py:241-316 and the teacher's CFG velocity of
seaweed_apt/generate.py:205-229 on the gfx950 path, as functions (the reference scripts' control flow — logging,
W&B, checkpoint cadence, dataset files — is not part of the hot path)."""
import importlib

import torch
import torch.nn.functional as F


@torch.no_grad()
def teacher_cfg_velocity(model, noise, t, context, context_null, guide_scale: float = 7.5, seq_len=None):
    """``v_teacher = v_uncond + guide_scale * (v_cond - v_uncond)`` for one latent (generate.py:205-229, BASELINE
    config 1): ``noise`` [C, F, H, W], ``t`` [1], ``context`` / ``context_null`` [L, text_dim].  The two forwards
    share x and t and run as ONE forward on a batch of two — the same kernels on twice the rows, bit-identical to
    two calls (16 vs 22 ms on the 1.3B model at S = 1560).  Returns fp32 [C_out, F, H, W]."""
    device = next(model.parameters()).device
    x = noise.to(device)
    if seq_len is None:
        p = model.patch_size
        seq_len = (x.shape[1] // p[0]) * (x.shape[2] // p[1]) * (x.shape[3] // p[2])
    tt = t.to(device).reshape(1)
    cond, uncond = model([x, x], torch.cat([tt, tt]), [context.to(device), context_null.to(device)], seq_len)
    return torch.add(uncond, cond - uncond, alpha=guide_scale)


def forward_backward(batch, distilled_model, num_train_timesteps=1000, gradient_accumulation_steps=1, loss_scale=1.0,
                     reference_loss_quirk=True):
    """The device side of ``training_step``: forward, loss, backward; returns the (accumulation-divided) loss as
    a 0-d device tensor without reading it back, so the whole function can be captured into a hipGraph
    (graphs.GraphedTrainingStep)."""
    noise, context, v_teacher = batch
    device = next(distilled_model.parameters()).device
    noise, context, v_teacher = noise.to(device), context.to(device), v_teacher.to(device)
    contexts_list = [context[i] for i in range(context.size(0))]
    patch = distilled_model.patch_size
    seq_len = (noise.shape[2] // patch[0]) * (noise.shape[3] // patch[1]) * (noise.shape[4] // patch[2])
    timestep = torch.ones(noise.shape[0], device=device) * num_train_timesteps
    out = distilled_model(noise, t=timestep, context=contexts_list, seq_len=seq_len)
    if reference_loss_quirk:
        v_student = out[0]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = F.mse_loss(v_student, v_teacher) / gradient_accumulation_steps
    else:
        loss = F.mse_loss(torch.stack(out), v_teacher) / gradient_accumulation_steps
    (loss * loss_scale).backward()
    return loss.detach()


def training_step(batch, distilled_model, num_train_timesteps=1000, gradient_accumulation_steps=1, loss_scale=1.0,
                  reference_loss_quirk=True):
    """batch = (noise [B,16,1,60,104], positive_contexts [B,512,4096], v_teacher [B,16,1,60,104]).

    Same arithmetic as the reference: seq_len from the patch size (distilled_trainer.py:261-264), t = 1000
    for every sample (:265), loss = mse(v_student, v_teacher) / accumulation steps, backward of the scaled
    loss (:289,301).  ``reference_loss_quirk`` keeps the reference's use of sample 0 only, broadcast against
    the whole teacher batch (:285-289); False uses every sample.  Returns the un-divided loss value."""
    loss = forward_backward(batch, distilled_model, num_train_timesteps, gradient_accumulation_steps, loss_scale,
                            reference_loss_quirk)
    return loss.item() * gradient_accumulation_steps
