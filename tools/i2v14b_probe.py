"""BASELINE config 5 on one MI355X: Wan2.1-I2V-14B backbone (d=5120, 40 heads, ffn 13824, 40 layers, in_dim 36,
257 CLIP tokens in the cross-attention) on an 81-frame 480x832 clip — VAE encode of the conditioning clip, K CFG
steps (2 forwards + fused UniPC update each), full VAE decode.  Random-init weights, synthetic inputs.
    python tools/i2v14b_probe.py [steps]
Prints one JSON object (kept under profiles/)."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "omnihuman-1-hack_amd"


def flops_14b_i2v(S, d=5120, f=13824, L=40, Lc=512 + 257):
    blk = 8 * S * d * d + 4 * S * S * d + (4 * S * d * d + 4 * Lc * d * d) + 4 * S * Lc * d + 4 * S * d * f
    return L * blk + 2 * 512 * (4096 * d + d * d) + 2 * S * (4 * 36) * d + 2 * S * d * 64


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda", 0)
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    i2v = importlib.import_module(PKG + ".wan.image2video")
    sched_mod = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    t0 = time.perf_counter()
    torch.manual_seed(5)
    with torch.device(dev):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.i2v_14B, model_type="i2v", in_dim=36))
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m = m.eval().requires_grad_(False)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    nparam = sum(p.numel() for p in m.parameters())
    g = torch.Generator(device=dev).manual_seed(9)
    F, lat_t, lat_h, lat_w = 81, 21, 60, 104
    S = lat_t * (lat_h // 2) * (lat_w // 2)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device=dev)
    clip = torch.cat([torch.rand(3, 1, 480, 832, device=dev, generator=g) * 2 - 1,
                      torch.zeros(3, F - 1, 480, 832, device=dev)], dim=1)
    vae.encode([clip[:, :5]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = vae.encode([clip])[0]
    torch.cuda.synchronize()
    enc_s = time.perf_counter() - t0
    y = torch.cat([i2v.first_frame_mask(F, lat_h, lat_w, device=dev), y])
    ctx = [torch.randn(120, 4096, device=dev, generator=g)]
    ctx_null = [torch.randn(40, 4096, device=dev, generator=g)]
    clip_fea = torch.randn(1, 257, 1280, device=dev, generator=g)
    st_c, st_u = m.encode_context(ctx, clip_fea=clip_fea), m.encode_context(ctx_null, clip_fea=clip_fea)
    sch = sched_mod.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(40, device=dev, shift=3.0)
    sch.set_begin_index(0)
    x = torch.randn(16, lat_t, lat_h, lat_w, device=dev, generator=g)

    def step(x):
        t = sch.timesteps[sch.step_index or 0].reshape(1).to(dev)
        c = m([x], t, st_c, S, y=[y])[0]
        u = m([x], t, st_u, S, y=[y])[0]
        return sch.step_cfg(c, u, 5.0, x)

    x = step(x)                                         # warm-up (packs 28.6 GB of bf16 weights)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        x = step(x)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    video = vae.decode([x])[0]
    torch.cuda.synchronize()
    dec_s = time.perf_counter() - t0
    fl = flops_14b_i2v(S)
    out = {"config": "Wan2.1-I2V-14B backbone, 81 frames 480x832 (latent [16,21,60,104] + 20 conditioning channels, "
                     f"S={S}), 257 CLIP + 512 text tokens, random-init weights, 1 MI355X",
           "params": nparam, "build_s": round(build_s, 1),
           "cfg_step_s": round(step_s, 3), "steps_per_s": round(1 / step_s, 4), "steps_timed": steps,
           "forward_tflop": round(fl / 1e12, 1), "achieved_tflops": round(2 * fl / step_s / 1e12, 1),
           "mfma_roofline_frac": round(2 * fl / step_s / 2.5e15, 4),
           "sample_40_steps_s_extrapolated": round(40 * step_s + enc_s + dec_s, 1),
           "vae_encode_s": round(enc_s, 3), "vae_decode_s": round(dec_s, 3),
           "video_shape": list(video.shape), "finite": bool(torch.isfinite(video).all() and torch.isfinite(x).all()),
           "hbm_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
