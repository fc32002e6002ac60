"""Launches only the FFN1-shaped GEMM (for rocprofv3 --pmc passes)."""
import importlib, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S, N, K = 32760, int(os.environ.get("N", 8960)), int(os.environ.get("K", 1536))
a = torch.randn(S, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(S, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(a, w, out=out, bias=bias, epilogue=ops.EPI_GELU_BF16 if os.environ.get("EPI", "gelu") == "gelu" else ops.EPI_BF16)
torch.cuda.synchronize()
