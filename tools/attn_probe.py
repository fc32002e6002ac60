"""Launches only the self-attention kernel at the benchmark shape (for rocprofv3 --pmc passes)."""
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
for _ in range(int(os.environ.get("N", 3))):
    ops.flash_attn(q, k, vt, None, out=o)
torch.cuda.synchronize()
