"""VAE parity at the benchmark's FULL area (480 x 832), against the CPU oracle: decode of a two-frame latent
[16,2,60,104] -> 5 pixel frames (the 'Rep' first chunk + one steady-state chunk through every temporal upsample) and
encode of those 5 frames back to 2 latent frames.  Minutes of host time, hence a probe and not a pytest case:
    python tests/probes/vae_full_area_parity.py > gpurun_out/vae_full_area_parity.json     (GPU box)
"""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wan_vae_oracle as V                                      # noqa: E402

PKG = "omnihuman-1-hack_amd"


def rel_rms(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30))


def main():
    dev = torch.device("cuda", 0)
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device=dev)
    sd = {k: v.detach().float().cpu() for k, v in vae.model.state_dict().items()}
    cfg = V.VAEConfig(dim=96)
    z = torch.randn(16, 2, 60, 104, generator=torch.Generator().manual_seed(5))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    ref = V.vae_decode(sd, cfg, z)
    t_dec = time.time() - t0
    out = vae.decode([z.to(dev)])[0].float().cpu()
    res = {"what": "480x832 (full benchmark area) VAE parity of the HIP path against oracle/wan_vae_oracle.py (fp32), random-init "
                   "weights of the Wan2.1 VAE architecture, seeded",
           "decode": {"latent": list(z.shape), "frames": list(ref.shape), "rel_rms": rel_rms(out, ref),
                      "max_abs_err": float((out - ref).abs().max()), "oracle_seconds": round(t_dec, 1)}}
    video = ref.clamp(-1, 1)
    t0 = time.time()
    mu_ref = V.vae_encode(sd, cfg, video)
    t_enc = time.time() - t0
    mu = vae.encode([video.to(dev)])[0].float().cpu()
    res["encode"] = {"frames": list(video.shape), "latent": list(mu_ref.shape), "rel_rms": rel_rms(mu, mu_ref),
                     "max_abs_err": float((mu - mu_ref).abs().max()), "oracle_seconds": round(t_enc, 1)}
    res["host_threads"] = torch.get_num_threads()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
