"""81-frame 480x832 encodes alone (for rocprofv3 --kernel-trace --stats).  Usage: python tools/vae_encode_only.py [bf16|fp32] [iters]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
it = int(sys.argv[2]) if len(sys.argv) > 2 else 2
vae = vae_mod.WanVAE(vae_pth=None, device="cuda", dtype=dt)
clip = torch.rand(3, 81, 480, 832, device="cuda") * 2 - 1
vae.encode([clip[:, :5]])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(it):
    lat = vae.encode([clip])[0]
torch.cuda.synchronize()
print("encode frames/s", 81 * it / (time.perf_counter() - t0), tuple(lat.shape))
