"""One warm-up + timed 81-frame 480x832 decodes and encodes (for rocprofv3 --kernel-trace --stats).
Usage: python tools/vae_decode_only.py [bf16|fp32] [iters]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
import bench
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
it = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print(vae_mod.bench_decode(torch.randn(16, 21, 60, 104, device="cuda"), "cuda", iters=it, telemetry=bench.Telemetry(0), dtype=dt))
