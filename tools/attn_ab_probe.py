"""ms per self-attention launch at the benchmark shape (S=32760, 12 heads), 20 launches after 5 warm-up; run once
per library (OMH_LIB) and alternate to A/B kernel revisions on one box."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S, H, D = int(os.environ.get("S", 32760)), 12, 128
q = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
k = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
Sp = (S + 63) // 64 * 64
vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
vt[:, :, :S] = torch.randn(1, H * D, S, device="cuda").to(torch.bfloat16)
o = torch.empty_like(q)
for _ in range(5):
    ops.flash_attn(q, k, vt, None, out=o)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.flash_attn(q, k, vt, None, out=o)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f"{os.environ.get('OMH_LIB', 'default')[-18:]} {ms:.4f} ms {4.0 * S * S * H * D / ms / 1e9:.1f} TF checksum {float(o.float().abs().sum()):.6e}")
