"""``WanT2V`` — the text-to-video pipeline surface of the reference
(seaweed_apt/wan/text2video.py:26-269) on the gfx950 DiT / VAE / sampler.

Same constructor arguments, attributes (``model, vae, text_encoder, vae_stride,
patch_size, num_train_timesteps, sample_neg_prompt, device, param_dtype``) and
``generate(...)`` signature.  The umT5 text encoder (``wan/modules/t5.py``, text2video.py:64-70) is built from
``checkpoint_dir`` when its checkpoint file is there (and ``disable_load_t5`` is not set); any callable
``text_encoder(list[str], device) -> list[Tensor[L, 4096]]`` can be plugged in instead, or pre-computed contexts
passed to ``generate``.

What changes under the hood (text2video.py:231-259): the two CFG forwards run
on the HIP DiT, the CFG combine + UniPC update is one fused kernel
(``FlowUniPCMultistepScheduler.step_cfg``), nothing is offloaded to the CPU
(the 1.3B/14B models fit one 288 GB MI355X), and the final latent goes through
the HIP VAE decoder.  FSDP / USP options of the reference are not built
(replicas only) and raise.
"""
import logging
import math
import os
import random
import sys
from typing import List, Optional

import torch
import torch.distributed as dist

from .modules.model import WanModel
from .modules.vae import WanVAE
from .utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .utils.fm_solvers_unipc import FlowUniPCMultistepScheduler


class WanT2V:

    def __init__(self, config, checkpoint_dir, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False, use_usp=False,
                 t5_cpu=False, disable_load_t5=False, text_encoder=None, model=None, vae=None):
        if t5_fsdp or dit_fsdp or use_usp:
            raise NotImplementedError("FSDP / USP sequence parallel are not part of this build: the DiT and VAE "
                                      "fit one MI355X, multi-GPU inference shards by clip (replicas)")
        self.device = torch.device(f"cuda:{device_id}")
        self.config = config
        self.rank = rank
        # t5_cpu (text2video.py:56, :205-214: keep umT5 on the host and move the context over) is accepted and has no
        # effect: the encoder runs on the HIP kernels only (no CPU path in this build) and 11 GB of bf16 umT5-XXL
        # next to a 1.3B / 14B DiT is nothing on 288 GB
        self.t5_cpu = t5_cpu
        self.num_train_timesteps = config.num_train_timesteps
        self.param_dtype = config.param_dtype
        if text_encoder is None and not disable_load_t5:
            ck = os.path.join(checkpoint_dir or "", config.t5_checkpoint)
            if checkpoint_dir and os.path.exists(ck):                  # text2video.py:64-70
                from .modules.t5 import T5EncoderModel
                text_encoder = T5EncoderModel(text_len=config.text_len, dtype=config.t5_dtype, device=self.device,
                                              checkpoint_path=ck,
                                              tokenizer_path=os.path.join(checkpoint_dir, config.t5_tokenizer))
        self.text_encoder = text_encoder
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
                               "WanT2V(text_encoder=callable)")
        return [t.to(self.device) for t in self.text_encoder(prompts, self.device)]

    def generate(self, input_prompt, size=(720, 512), frame_num=81, shift=5.0, sample_solver="unipc",
                 sampling_steps=50, guide_scale=5.0, n_prompt="", seed=-1, offload_model=True,
                 context: Optional[List[torch.Tensor]] = None, context_null: Optional[List[torch.Tensor]] = None,
                 return_latent: bool = False, batched_cfg: bool = True, cfg_split=None):
        r"""text2video.py:112-269.  Returns the video ``[3, N, H, W]`` on rank 0 (else None).

        ``cfg_split``: a ``parallel.CFGPairSplit`` — this clip's two CFG branches run on two GPUs, one all-gather
        of the velocity predictions per step; every rank of the pair passes the same arguments."""
        F = frame_num
        target_shape = (self.vae.model.z_dim, (F - 1) // self.vae_stride[0] + 1, size[1] // self.vae_stride[1],
                        size[0] // self.vae_stride[2])
        seq_len = math.ceil((target_shape[2] * target_shape[3]) / (self.patch_size[1] * self.patch_size[2]) *
                            target_shape[1] / self.sp_size) * self.sp_size
        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        seed = seed if seed >= 0 else random.randint(0, sys.maxsize)
        if cfg_split is not None:
            seed = cfg_split.sync_seed(seed)
        seed_g = torch.Generator(device=self.device)
        seed_g.manual_seed(seed)
        if context is None:
            context = self._encode([input_prompt])
        if context_null is None:
            context_null = self._encode([n_prompt])
        context = [t.to(self.device) for t in context]
        context_null = [t.to(self.device) for t in context_null]
        noise = [torch.randn(*target_shape, dtype=torch.float32, device=self.device, generator=seed_g)]

        if sample_solver not in ("unipc", "dpm++"):
            raise NotImplementedError("Unsupported solver.")
        with torch.no_grad():
            if sample_solver == "unipc":                       # text2video.py:204-211
                sample_scheduler = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                               use_dynamic_shifting=False)
                sample_scheduler.set_timesteps(sampling_steps, device=self.device, shift=shift)
                timesteps = sample_scheduler.timesteps
            else:                                              # text2video.py:212-221
                sample_scheduler = FlowDPMSolverMultistepScheduler(num_train_timesteps=self.num_train_timesteps,
                                                                   shift=1, use_dynamic_shifting=False)
                timesteps, _ = retrieve_timesteps(sample_scheduler, device=self.device,
                                                  sigmas=get_sampling_sigmas(sampling_steps, shift))
            sample_scheduler.set_begin_index(0)
            latents = noise
            # text embedding + per-block cross-attention K/V do not depend on (x, t): once per sample, not 100x
            # The conditional and unconditional forwards of a step share x and t: run them as ONE forward on a
            # batch of two (same kernels on twice the rows, bit-identical outputs) unless that would push a GEMM
            # operand past the kernels' 2 GiB limit (14B at 720p); ``batched_cfg=False`` keeps two calls.
            batched = batched_cfg and 2 * (seq_len + 128) * getattr(self.model, "ffn_dim", 0) * 2 < 0x7fffffff
            # Long sequences (every launch fills the chip with one clip's rows): the two forwards as a PAIR that computes
            # what they share — embeddings and block 0's self-attention sub-layer, whose inputs are x and t alone — once
            # (WanModel.forward_cfg_pair: the same bits as two calls, one self-attention launch of 60 less per step)
            pair = cfg_split is None and seq_len >= 8192
            if cfg_split is not None:
                batched = False
                mine = self.model.encode_context(context if cfg_split.runs_conditional else context_null)
            elif pair:
                batched = False
                context, context_null = self.model.encode_context(context), self.model.encode_context(context_null)
            elif batched:
                both = self.model.encode_context([context[0], context_null[0]])
            else:
                context, context_null = self.model.encode_context(context), self.model.encode_context(context_null)
            for t in timesteps:
                if cfg_split is not None:
                    cond, uncond = cfg_split.exchange(
                        self.model(latents, t=torch.stack([t]), context=mine, seq_len=seq_len)[0])
                elif pair:
                    cond, uncond = self.model.forward_cfg_pair(latents, torch.stack([t]), context, context_null, seq_len)
                    cond, uncond = cond[0], uncond[0]
                elif batched:
                    cond, uncond = self.model([latents[0], latents[0]], t=torch.stack([t, t]), context=both,
                                              seq_len=seq_len)
                else:
                    timestep = torch.stack([t])
                    cond = self.model(latents, t=timestep, context=context, seq_len=seq_len)[0]
                    uncond = self.model(latents, t=timestep, context=context_null, seq_len=seq_len)[0]
                # noise_pred = uncond + g (cond - uncond); latents = scheduler.step(noise_pred, t, latents)
                latents = [sample_scheduler.step_cfg(cond, uncond, guide_scale, latents[0])]
            x0 = latents
            videos = None
            if self.rank == 0:
                videos = x0 if return_latent else self.vae.decode(x0)
        if dist.is_initialized():
            dist.barrier()
        return videos[0] if self.rank == 0 else None
