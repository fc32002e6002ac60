"""One warm-up + two timed 81-frame 480x832 decodes and encodes (for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
import bench
print(vae_mod.bench_decode(torch.randn(16, 21, 60, 104, device="cuda"), "cuda", iters=2, telemetry=bench.Telemetry(0)))
