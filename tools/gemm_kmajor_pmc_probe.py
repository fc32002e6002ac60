"""A few launches each of the row-major-B, k-major-B and TN GEMM for rocprofv3 --pmc (LDS bank conflicts)."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
ops.set_option("OMH_GEMM_TILE", "big")
ops.set_option("OMH_GEMM_TN_TILE", "big")
ops.set_option("OMH_GEMM_TN_SPLIT", "1")
M, N, K = 6240, 1536, 8960
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(K, N, device="cuda") / math.sqrt(K)).bfloat16()
wt = w.t().contiguous()
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.gemm(a, w, out=out, epilogue=ops.EPI_F32, b_kmajor=True)
    ops.gemm(a, wt, out=out, epilogue=ops.EPI_F32)
x = torch.randn(6240, 1536, device="cuda").bfloat16()
dy = torch.randn(6240, 8960, device="cuda").bfloat16()
for _ in range(3):
    ops.gemm_tn(dy, x)
torch.cuda.synchronize()
