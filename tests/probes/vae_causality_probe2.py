"""Is the T=4 decode as close to the oracle as the T<=3 decodes? (quarter area, dim=96)"""
import importlib, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import wan_vae_oracle as V
vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
torch.manual_seed(4321)
vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")
sd = {k: v.detach().float().cpu() for k, v in vae.model.state_dict().items()}
cfg = V.VAEConfig(dim=96)
torch.set_num_threads(32)
H, W = int(sys.argv[1]), int(sys.argv[2])
z = torch.randn(16, 5, H, W, generator=torch.Generator().manual_seed(5))
ref = V.vae_decode(sd, cfg, z)
res = {}
for n in (1, 2, 3, 4, 5):
    out = vae.decode([z[:, :n].cuda().contiguous()])[0].float().cpu()
    F = out.shape[1]
    res[f"T{n}_vs_oracle"] = [round(float((out[:, f] - ref[:, f]).norm() / ref[:, f].norm()), 5) for f in range(F)]
print(json.dumps(res))
