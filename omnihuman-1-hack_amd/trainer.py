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
    a 0-d device tensor without reading it back (no host synchronisation inside a step)."""
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


# ----------------------------------------------------------------------------------------------------------------
# Checkpoint save / resume in the trainers' formats (seaweed_apt/distilled_trainer.py:153-178, 199-231): the
# ``accelerator.save_state`` directory layout (model.safetensors, optimizer.bin, scaler.pt, random_states_<rank>.pkl)
# OR the reference's manual fallback file (pytorch_model.bin = {'model', 'optimizer', 'scaler', 'step',
# 'epoch'}), and the EMA weights as a plain state dict (ema_model_step_<n>.pt / ema_model_epoch_<n>.pt /
# ema_model_final.pt, read back by eval_ema.py:43-47 and wan_inference.py:59-63).  Host-side I/O only.
def _numpy_safe_globals():
    """What a numpy RNG state needs under torch.load(weights_only=True) (accelerate writes random_states_<rank>.pkl
    with torch.save: python tuples, one uint32 ndarray, torch ByteTensors)."""
    import numpy as np
    core = getattr(np, "_core", None) or np.core
    return [core.multiarray._reconstruct, np.ndarray, np.dtype, type(np.dtype(np.uint32))]


def save_checkpoint(checkpoint_dir, model, optimizer=None, scaler_state=None, step: int = 0, epoch: int = 0, rank: int = 0,
                    is_main_process: bool = True, manual_fallback: bool = False):
    """Write ``checkpoint_dir`` in ONE of the reference's two layouts (distilled_trainer.py:153-178): the
    ``accelerator.save_state`` directory (model.safetensors, optimizer.bin, scaler.pt, random_states_<rank>.pkl; step and
    epoch in the small side file train_state.json) or, with ``manual_fallback``, the single pytorch_model.bin =
    {'model', 'optimizer', 'scaler', 'step', 'epoch'} the reference writes when save_state fails.  The model is written
    once.  Every rank writes its RNG states (as accelerate does); the main process writes the rest."""
    import json
    import os
    import random
    import numpy as np
    os.makedirs(checkpoint_dir, exist_ok=True)
    states = {"step": int(step), "random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
              "torch_manual_seed": torch.get_rng_state()}
    if torch.cuda.is_available():
        states["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    torch.save(states, os.path.join(checkpoint_dir, f"random_states_{rank}.pkl"))
    if not is_main_process:
        return checkpoint_dir
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    opt_sd = optimizer.state_dict() if optimizer is not None else None
    if manual_fallback:
        torch.save({"model": sd, "optimizer": opt_sd, "scaler": scaler_state, "step": int(step), "epoch": int(epoch)},
                   os.path.join(checkpoint_dir, "pytorch_model.bin"))
        return checkpoint_dir
    try:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(checkpoint_dir, "model.safetensors"), metadata={"format": "pt"})
    except (ImportError, RuntimeError):                       # no safetensors / tied parameters: accelerate's fallback file
        torch.save(sd, os.path.join(checkpoint_dir, "pytorch_model.bin"))
    if opt_sd is not None:
        torch.save(opt_sd, os.path.join(checkpoint_dir, "optimizer.bin"))
    if scaler_state is not None:
        torch.save(scaler_state, os.path.join(checkpoint_dir, "scaler.pt"))
    with open(os.path.join(checkpoint_dir, "train_state.json"), "w") as fh:
        json.dump({"step": int(step), "epoch": int(epoch)}, fh)
    return checkpoint_dir


def load_checkpoint(checkpoint_dir, model, optimizer=None, rank: int = 0, restore_rng: bool = True):
    """Resume from a directory written by ``save_checkpoint``, by ``accelerator.save_state`` or by the reference's
    manual fallback.  Returns ``{'step', 'epoch', 'scaler', 'rng_restored'}`` (what the files hold of them).  A
    torch.optim.AdamW state (tensor ``step`` entries) loads into ``optim.AdamW`` as is.  Every file is read with
    ``weights_only=True`` (tensors and plain containers; the RNG file additionally numpy's array types): a checkpoint
    directory is data, not code.  pytorch_model.bin is opened only when something is missing without it."""
    import json
    import os
    import random
    import numpy as np
    info = {"step": 0, "epoch": 0, "scaler": None, "rng_restored": False}
    load = lambda path: torch.load(path, map_location="cpu", weights_only=True)
    manual = os.path.join(checkpoint_dir, "pytorch_model.bin")
    st_path = os.path.join(checkpoint_dir, "model.safetensors")
    opt_path = os.path.join(checkpoint_dir, "optimizer.bin")
    side = os.path.join(checkpoint_dir, "train_state.json")
    blob = None

    def manual_blob():
        nonlocal blob
        if blob is None and os.path.exists(manual):
            blob = load(manual)
            if not (isinstance(blob, dict) and "model" in blob):  # accelerate's pytorch_model.bin is the bare state dict
                blob = {"model": blob}
        return blob or {}

    if os.path.exists(st_path):
        from safetensors.torch import load_file
        model.load_state_dict(load_file(st_path))
    elif "model" in manual_blob():
        model.load_state_dict(blob["model"])
    else:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {checkpoint_dir}")
    if optimizer is not None:
        opt_sd = load(opt_path) if os.path.exists(opt_path) else manual_blob().get("optimizer")
        if opt_sd is not None:
            optimizer.load_state_dict(opt_sd)
    sc = os.path.join(checkpoint_dir, "scaler.pt")
    info["scaler"] = load(sc) if os.path.exists(sc) else (blob or {}).get("scaler")
    have_step = False
    if os.path.exists(side):
        with open(side) as fh:
            st = json.load(fh)
        info["step"], info["epoch"] = int(st.get("step", 0)), int(st.get("epoch", 0))
        have_step = "step" in st
    elif "step" in manual_blob():
        info["step"], info["epoch"] = int(blob.get("step", 0)), int(blob.get("epoch", 0))
        have_step = True
    rs = os.path.join(checkpoint_dir, f"random_states_{rank}.pkl")
    if os.path.exists(rs):
        try:
            with torch.serialization.safe_globals(_numpy_safe_globals()):
                states = load(rs)
        except Exception as exc:                                  # not a torch.save file / types outside the allow-list
            import warnings
            warnings.warn(f"{rs}: RNG states not restored ({type(exc).__name__})")
            states = None
        if states is not None:
            # accelerate keeps its own micro-step counter in this file: it is the trainer's step only when nothing
            # else recorded one (train_state.json / the manual blob win)
            if not have_step:
                info["step"] = int(states.get("step", info["step"]))
            if restore_rng:
                random.setstate(states["random_state"])
                np.random.set_state(states["numpy_random_seed"])
                torch.set_rng_state(states["torch_manual_seed"])
                if torch.cuda.is_available() and "torch_cuda_manual_seed" in states:
                    try:
                        torch.cuda.set_rng_state_all(states["torch_cuda_manual_seed"])
                    except (RuntimeError, IndexError):            # saved on a different number of devices
                        pass
                info["rng_restored"] = True
    # parameters were rewritten: the packed bf16 copies are keyed on the parameters' versions (load_state_dict's copy_
    # bumps them), nothing else to invalidate
    return info


def save_ema(path, ema_model):
    """distilled_trainer.py:175-178, 223-231: the EMA weights as a bare state dict."""
    torch.save({k: v.detach().cpu() for k, v in ema_model.state_dict().items()}, path)
    return path


def load_ema(path, model):
    """eval_ema.py:43-47 / wan_inference.py:59-63."""
    model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return model
