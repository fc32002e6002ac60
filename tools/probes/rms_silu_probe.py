"""rms_silu_cl at the VAE decoder's shapes: us per call and HBM GB/s (read + write).  GPU box."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
out = []
for (P, C) in ((8 * 480 * 832, 96), (8 * 240 * 416, 192), (4 * 120 * 208, 384), (2 * 60 * 104, 384)):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(P, C, device="cuda").to(dt)
        g = torch.rand(C, device="cuda") + 0.5
        y = torch.empty(P, C, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.rms_silu_cl(x, g, out=y)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.rms_silu_cl(x, g, out=y)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        by = P * C * (x.element_size() + 2)
        out.append({"P": P, "C": C, "in": str(dt).split(".")[1], "us": round(us, 1), "GBps": round(by / us / 1e3, 1)})
print(json.dumps(out, indent=0))
