"""fp32 residual trunk vs the round-1 bf16 trunk of the VAE executor (OMH_VAE_TRUNK): parity against the fp32 oracle on
the wide-tile decode ([16,2,30,52] -> 5 frames 240x416) and a 9-frame encode, and time of the full 81-frame 480x832
decode / encode.  Each mode in its own process (the switch is read at import)."""
import importlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child():
    import torch
    from oracle import wan_vae_oracle as V, detgen
    vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    cfg = V.VAEConfig(dim=96)
    sd = V.synth_state_dict(cfg, "vae96wide")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=96)
    vae.model.load_state_dict(sd)
    z = torch.from_numpy(detgen.normalish("vae/zwide", (16, 4, 16, 24)))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0]
    vid = ref.clamp(-1, 1)
    refe = V.vae_encode(sd, cfg, vid)
    oute = vae.encode([vid.cuda()])[0]
    print(f"trunk {os.environ.get('OMH_VAE_TRUNK', 'f32')}: decode rel-RMS {rel(out, ref):.3e}  encode rel-RMS {rel(oute, refe):.3e}", flush=True)
    res = vae_mod.bench_decode(torch.randn(16, 21, 60, 104, device="cuda"), "cuda", iters=2)
    print(f"trunk {os.environ.get('OMH_VAE_TRUNK', 'f32')}: {res}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for g2 in ("1", "2"):
            print("== group2", g2, flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OMH_VAE_GROUP2=g2), check=False)
