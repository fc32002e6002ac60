"""BASELINE config 4 on one MI355X: OmniHumanWanT2V on the Wan2.1-T2V-1.3B backbone — 49 pixel frames 480x832
(13 latent frames [16,13,60,104]) + 1 reference latent frame concatenated along T (S = 14 * 1560 = 21 840 tokens),
wav2vec-sized audio features [1,49,1024], pose heatmaps [1,K,49,64,64].  Times the adapters and
the annealed-CFG denoising step; random-init weights of the real architecture, synthetic inputs."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "omnihuman-1-hack_amd"
omni = importlib.import_module(PKG + ".omnihuman_wan_t2v")
model_mod = importlib.import_module(PKG + ".wan.modules.model")
cfgs = importlib.import_module(PKG + ".wan.configs")
vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
K = int(os.environ.get("OMH_POSE_KEYPOINTS", "308"))           # omni_config.yaml: num_keypoints 308
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.manual_seed(1)
with torch.device(dev):
    dit = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.t2v_1_3B))
    torch.nn.init.xavier_uniform_(dit.head.head.weight)
dit.eval().requires_grad_(False)
vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device=dev)


class T2V:
    model, text_encoder = dit, None


T2V.vae = vae
m = omni.OmniHumanWanT2V(dict(num_frames=49, num_keypoints=K, model_dim=1536, audio_dim=1024), device_id=0, wan_t2v=T2V)
g = torch.Generator(device=dev).manual_seed(2)
audio = torch.randn(1, 49, 1024, device=dev, generator=g)
pose = torch.rand(1, K, 49, 64, 64, device=dev, generator=g)
ref_img = torch.rand(3, 1, 480, 832, device=dev, generator=g) * 2 - 1
ctx = torch.randn(120, 4096, device=dev, generator=g)
ctx0 = torch.randn(40, 4096, device=dev, generator=g)
noise = torch.randn(16, 13, 60, 104, device=dev, generator=g)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


res = {"workload": f"OmniHumanWanT2V on Wan2.1-T2V-1.3B: latent [16,13,60,104] + reference frame (S=21840), audio "
                   f"[1,49,1024], pose [1,{K},49,64,64]"}
ta, a = timed(lambda: m.process_audio(audio))
tp, p = timed(lambda: m.process_pose(pose))
tt_, tok = timed(lambda: m.condition_tokens(a, p))
tr, ref = timed(lambda: m.process_reference(ref_img), 3)
res.update(audio_adapter_ms=round(ta * 1e3, 3), pose_adapter_ms=round(tp * 1e3, 3), condition_tokens_ms=round(tt_ * 1e3, 3),
           reference_encode_ms=round(tr * 1e3, 2), condition_tokens=list(tok.shape))
pose_flops = 2 * 27 * 49 * (64 * 64 * K * 128 + 32 * 32 * 128 * 256 + 16 * 16 * 256 * 384) + 2 * 49 * 384 * 256 * 1536
res["pose_adapter_tflops"] = round(pose_flops / tp / 1e12, 1)
# the sampling loop (forward() = prepare_conditions + N steps + decode): time N and 1 step, difference = per step
kw = dict(audio=audio, pose=pose, reference_image=ref_img, cfg_scale=7.5, text_context=ctx, text_context_null=ctx0,
          noise=noise, return_latent=True)
t1, _ = timed(lambda: m(num_inference_steps=1, **kw), 2)
tn, lat = timed(lambda: m(num_inference_steps=1 + STEPS, **kw), 2)
step = (tn - t1) / STEPS
S = 14 * 1560
d, f, L, Lc = 1536, 8960, 30, 512
F_block = 8 * S * d * d + 4 * S * S * d + (4 * S * d * d) + 4 * S * Lc * d + 4 * S * d * f     # per-step K/V of the context cached
res.update(step_ms=round(step * 1e3, 1), steps_per_s=round(1 / step, 3),
           achieved_tflops=round(2 * L * F_block / step / 1e12, 1), mfma_roofline_frac=round(2 * L * F_block / step / 2.5e15, 4),
           one_step_sample_incl_conditions_ms=round(t1 * 1e3, 1), finite=bool(torch.isfinite(lat).all()))
td, vid = timed(lambda: vae.decode([lat])[0], 2)
res.update(decode_49_frames_ms=round(td * 1e3, 1), video=list(vid.shape))
print(json.dumps(res))
