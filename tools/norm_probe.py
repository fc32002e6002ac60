"""rmsnorm_rope / layernorm_modulate launch times at S = 32760, d = 1536 (HBM-bound kernels of the DiT block)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S, d, D = 32760, 1536, 128
qk = torch.randn(S, 2 * d, device="cuda").bfloat16()
out = torch.empty(S, d, device="cuda", dtype=torch.bfloat16)
w = torch.rand(d, device="cuda") + 0.5
cos = torch.randn(1024, D // 2, device="cuda"); sin = torch.randn(1024, D // 2, device="cuda")
grid = torch.tensor([21, 30, 52], dtype=torch.int32, device="cuda")
x = torch.randn(S, d, device="cuda")


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


f1 = lambda: ops.rmsnorm_rope_bf16_raw(ops.ptr(qk), 2 * d, ops.ptr(out), S, d, ops.ptr(w), 1e-6, 1, ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), S, out_scale=0.1275)
us = t(f1)
print(f"rmsnorm_rope bf16 (q of q|k): {us:.1f} us  {S * d * 4 / us / 1e6:.2f} TB/s")
e1 = torch.randn(1, 6, d, device="cuda")
f2 = lambda: ops.layernorm_modulate(x, 1e-6, 1.0, mul1=e1[:, 1], add1=e1[:, 0], rows_per_batch=S)     # as the DiT block calls it
us = t(f2)
print(f"layernorm_modulate: {us:.1f} us  {S * d * 6 / us / 1e6:.2f} TB/s")
