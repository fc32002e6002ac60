"""Prefix-vs-whole decode/encode at 480x832: per-frame relative RMS difference (rounding noise only if causal)."""
import importlib, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
torch.manual_seed(4321)
vae = vae_mod.WanVAE(vae_pth=None, device="cuda")
g = torch.Generator(device="cuda").manual_seed(6)
z = torch.randn(16, 4, 60, 104, device="cuda", generator=g)
full = vae.decode([z])[0]
again = vae.decode([z])[0]
res = {"decode_repeat_equal": bool(torch.equal(full, again))}
for n in (1, 2, 3):
    head = vae.decode([z[:, :n].contiguous()])[0]
    F = head.shape[1]
    res[f"decode_prefix{n}"] = [float((head[:, f] - full[:, f]).norm() / full[:, f].norm()) for f in range(F)]
video = full.clamp(-1, 1)
mu = vae.encode([video])[0]
for n in (1, 5, 9):
    m = vae.encode([video[:, :n].contiguous()])[0]
    res[f"encode_prefix{n}"] = [float((m[:, f] - mu[:, f]).norm() / mu[:, f].norm()) for f in range(m.shape[1])]
print(json.dumps(res, indent=1))
