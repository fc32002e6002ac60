"""The student training step of seaweed_apt/distilled_trainer.py:241-316 on the
gfx950 path, as a function (the reference trainer's control flow — logging,
W&B, checkpoint cadence — is not part of the hot path)."""
import importlib

import torch
import torch.nn.functional as F


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
