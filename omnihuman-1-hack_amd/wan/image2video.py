"""``WanI2V`` — the image-to-video pipeline surface of the reference
(seaweed_apt/wan/image2video.py:29-347) on the gfx950 DiT / VAE / sampler.

Same constructor arguments, attributes and ``generate(...)`` signature.  The
umT5 text encoder and the CLIP ViT-H image encoder (``wan/modules/t5.py``,
``wan/modules/clip.py``; image2video.py:74-92) are built from ``checkpoint_dir``
when their checkpoint files are there; otherwise ``text_encoder`` may be any
callable ``(list[str], device) -> list[Tensor[L, 4096]]`` and ``clip`` any
object with ``visual(list[Tensor[3,1,H,W]]) -> Tensor[1, 257, 1280]``, or pass
``context= / context_null= / clip_fea=`` to ``generate``.

What this pipeline does on the device (image2video.py:186-331): VAE-encode the
conditioning clip (first frame = the bicubic-resized image, the rest zeros),
build the 4-channel first-frame mask, then per step two DiT forwards on
``cat(latent, mask, y)`` with the CLIP tokens in the cross-attention, the fused
CFG + scheduler kernel, and finally the VAE decode.  Everything derived from
(context, clip_fea) alone is computed once per sample (``WanModel.encode_context``).
"""
import logging
import math
import os
import random
import sys
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .modules.model import WanModel
from .modules.vae import WanVAE
from .utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

__all__ = ["WanI2V", "first_frame_mask"]


def _to_tensor(img) -> torch.Tensor:
    """torchvision's ``to_tensor`` for the two cases the pipeline meets: a PIL image (HWC uint8 -> CHW [0,1])
    or an already-CHW float tensor in [0, 1]."""
    if isinstance(img, torch.Tensor):
        return img.detach().to(torch.float32).clone()
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.to(torch.float32).div(255.0) if arr.dtype == np.uint8 else t.to(torch.float32)


def first_frame_mask(frame_num: int, lat_h: int, lat_w: int, device=None) -> torch.Tensor:
    """image2video.py:203-210: 1 on the conditioning frame, 0 elsewhere, folded 4 pixel frames -> 1 latent
    frame as channels: ``[4, (frame_num-1)//4+1, lat_h, lat_w]``."""
    msk = torch.ones(1, frame_num, lat_h, lat_w, device=device)
    msk[:, 1:] = 0
    msk = torch.concat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
    msk = msk.view(1, msk.shape[1] // 4, 4, lat_h, lat_w)
    return msk.transpose(1, 2)[0]


class WanI2V:

    def __init__(self, config, checkpoint_dir, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False, use_usp=False,
                 t5_cpu=False, init_on_cpu=True, text_encoder=None, clip=None, model=None, vae=None):
        if t5_fsdp or dit_fsdp or use_usp:
            raise NotImplementedError("FSDP / USP sequence parallel are not part of this build: the DiT and VAE "
                                      "fit one MI355X, multi-GPU inference shards by clip (replicas)")
        self.device = torch.device(f"cuda:{device_id}")
        self.config = config
        self.rank = rank
        self.use_usp = False
        # t5_cpu (text2video.py:56, :205-214: keep umT5 on the host and move the context over) is accepted and has no
        # effect: the encoder runs on the HIP kernels only (no CPU path in this build) and 11 GB of bf16 umT5-XXL
        # next to a 1.3B / 14B DiT is nothing on 288 GB
        self.t5_cpu = t5_cpu
        self.num_train_timesteps = config.num_train_timesteps
        self.param_dtype = config.param_dtype
        if text_encoder is None:
            ck = os.path.join(checkpoint_dir or "", config.t5_checkpoint)
            if checkpoint_dir and os.path.exists(ck):                  # image2video.py:74-81
                from .modules.t5 import T5EncoderModel
                text_encoder = T5EncoderModel(text_len=config.text_len, dtype=config.t5_dtype, device=self.device,
                                              checkpoint_path=ck,
                                              tokenizer_path=os.path.join(checkpoint_dir, config.t5_tokenizer))
        if clip is None:
            ck = os.path.join(checkpoint_dir or "", config.clip_checkpoint)
            if checkpoint_dir and os.path.exists(ck):                  # image2video.py:87-92
                from .modules.clip import CLIPModel
                clip = CLIPModel(dtype=config.clip_dtype, device=self.device, checkpoint_path=ck,
                                 tokenizer_path=os.path.join(checkpoint_dir, config.clip_tokenizer))
        self.text_encoder = text_encoder
        self.clip = clip
        self.vae_stride = config.vae_stride
        self.patch_size = config.patch_size
        if vae is None:
            vae = WanVAE(vae_pth=os.path.join(checkpoint_dir, config.vae_checkpoint), device=self.device,
                         dtype=getattr(config, "vae_dtype", torch.float))     # the reference's arithmetic (text2video.py:81-83, vae.py:619-624); config.vae_dtype = torch.bfloat16 opts into bf16 operands
        self.vae = vae
        if model is None:
            logging.info(f"Creating WanModel from {checkpoint_dir}")
            model = WanModel.from_pretrained(checkpoint_dir)
        self.model = model
        self.model.eval().requires_grad_(False)
        self.sp_size = 1
        if dist.is_initialized():
            dist.barrier()
        self.model.to(self.device)
        self.sample_neg_prompt = config.sample_neg_prompt

    def _encode(self, prompts: List[str]):
        if self.text_encoder is None:
            raise RuntimeError("no text encoder attached: pass context=/context_null= to generate(), or construct "
                               "WanI2V(text_encoder=callable)")
        return [t.to(self.device) for t in self.text_encoder(prompts, self.device)]

    def generate(self, input_prompt, img, max_area=720 * 1280, frame_num=81, shift=5.0, sample_solver="unipc",
                 sampling_steps=40, guide_scale=5.0, n_prompt="", seed=-1, offload_model=True,
                 context: Optional[List[torch.Tensor]] = None, context_null: Optional[List[torch.Tensor]] = None,
                 clip_fea: Optional[torch.Tensor] = None, return_latent: bool = False, batched_cfg: bool = True,
                 cfg_split=None):
        r"""image2video.py:129-347.  ``img``: PIL image or float tensor [3, H, W] in [0, 1].  Returns the video
        ``[3, N, H, W]`` on rank 0 (else None).  (The reference hard-codes 21 latent / 81 pixel frames in the
        noise and mask shapes, :196-203; here they follow ``frame_num`` and coincide at the default 81.)
        ``cfg_split``: a ``parallel.CFGPairSplit`` — the two CFG branches of this clip on two GPUs (see WanT2V)."""
        img = _to_tensor(img).sub_(0.5).div_(0.5).to(self.device)
        F = frame_num
        T_lat = (F - 1) // self.vae_stride[0] + 1
        h, w = img.shape[1:]
        aspect_ratio = h / w
        lat_h = round(np.sqrt(max_area * aspect_ratio) // self.vae_stride[1] // self.patch_size[1] *
                      self.patch_size[1])
        lat_w = round(np.sqrt(max_area / aspect_ratio) // self.vae_stride[2] // self.patch_size[2] *
                      self.patch_size[2])
        h = lat_h * self.vae_stride[1]
        w = lat_w * self.vae_stride[2]
        max_seq_len = T_lat * lat_h * lat_w // (self.patch_size[1] * self.patch_size[2])
        max_seq_len = int(math.ceil(max_seq_len / self.sp_size)) * self.sp_size

        seed = seed if seed >= 0 else random.randint(0, sys.maxsize)
        if cfg_split is not None:
            seed = cfg_split.sync_seed(seed)
        seed_g = torch.Generator(device=self.device)
        seed_g.manual_seed(seed)
        noise = torch.randn(16, T_lat, lat_h, lat_w, dtype=torch.float32, generator=seed_g, device=self.device)
        msk = first_frame_mask(F, lat_h, lat_w, device=self.device)

        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        if context is None:
            context = self._encode([input_prompt])
        if context_null is None:
            context_null = self._encode([n_prompt])
        context = [t.to(self.device) for t in context]
        context_null = [t.to(self.device) for t in context_null]
        if clip_fea is None:
            if self.clip is None:
                raise RuntimeError("no CLIP image encoder attached: pass clip_fea= to generate(), or construct "
                                   "WanI2V(clip=object with .visual())")
            clip_fea = self.clip.visual([img[:, None, :, :]])
        clip_context = clip_fea.to(self.device)

        # conditioning clip -> latent (image2video.py:226-236); the resize runs on the host like the reference's
        first = torch.nn.functional.interpolate(img[None].cpu(), size=(h, w), mode="bicubic").transpose(0, 1)
        y = self.vae.encode([torch.concat([first, torch.zeros(3, F - 1, h, w)], dim=1).to(self.device)])[0]
        y = torch.concat([msk, y])

        if sample_solver not in ("unipc", "dpm++"):
            raise NotImplementedError("Unsupported solver.")
        with torch.no_grad():
            if sample_solver == "unipc":                       # image2video.py:247-254
                sample_scheduler = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                               use_dynamic_shifting=False)
                sample_scheduler.set_timesteps(sampling_steps, device=self.device, shift=shift)
                timesteps = sample_scheduler.timesteps
            else:                                              # image2video.py:255-264
                sample_scheduler = FlowDPMSolverMultistepScheduler(num_train_timesteps=self.num_train_timesteps,
                                                                   shift=1, use_dynamic_shifting=False)
                timesteps, _ = retrieve_timesteps(sample_scheduler, device=self.device,
                                                  sigmas=get_sampling_sigmas(sampling_steps, shift))
            sample_scheduler.set_begin_index(0)
            latent = noise
            # text/image embedding and per-block cross-attention K/V (text and image tokens): once per sample
            # cond / uncond as one forward on a batch of two where the operands stay below the kernels' 2 GiB limit
            # (see WanT2V.generate); bit-identical to two calls
            batched = batched_cfg and 2 * (max_seq_len + 128) * getattr(self.model, "ffn_dim", 0) * 2 < 0x7fffffff
            # long sequences: the pair that shares block 0's self-attention sub-layer (see WanT2V.generate)
            pair = cfg_split is None and max_seq_len >= 8192
            if cfg_split is not None:
                batched = False
                mine = self.model.encode_context([context[0]] if cfg_split.runs_conditional else context_null,
                                                 clip_fea=clip_context)
            elif pair:
                batched = False
                arg_c = self.model.encode_context([context[0]], clip_fea=clip_context)
                arg_null = self.model.encode_context(context_null, clip_fea=clip_context)
            elif batched:
                both = self.model.encode_context([context[0], context_null[0]],
                                                 clip_fea=torch.cat([clip_context, clip_context]))
            else:
                arg_c = self.model.encode_context([context[0]], clip_fea=clip_context)
                arg_null = self.model.encode_context(context_null, clip_fea=clip_context)
            for t in timesteps:
                if cfg_split is not None:
                    cond, uncond = cfg_split.exchange(self.model(
                        [latent], t=torch.stack([t]).to(self.device), context=mine, seq_len=max_seq_len, y=[y])[0])
                elif pair:
                    cond, uncond = self.model.forward_cfg_pair([latent], torch.stack([t]).to(self.device), arg_c, arg_null,
                                                               max_seq_len, y=[y])
                    cond, uncond = cond[0], uncond[0]
                elif batched:
                    cond, uncond = self.model([latent, latent], t=torch.stack([t, t]).to(self.device), context=both,
                                              seq_len=max_seq_len, y=[y, y])
                else:
                    timestep = torch.stack([t]).to(self.device)
                    cond = self.model([latent], t=timestep, context=arg_c, seq_len=max_seq_len, y=[y])[0]
                    uncond = self.model([latent], t=timestep, context=arg_null, seq_len=max_seq_len, y=[y])[0]
                latent = sample_scheduler.step_cfg(cond, uncond, guide_scale, latent)
            x0 = [latent]
            videos = None
            if self.rank == 0:
                videos = x0 if return_latent else self.vae.decode(x0)
        if dist.is_initialized():
            dist.barrier()
        return videos[0] if self.rank == 0 else None
