"""Launches only the FFN1-shaped GEMM (for rocprofv3 --pmc passes)."""
import importlib, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S, N, K = 32760, int(os.environ.get("N", 8960)), int(os.environ.get("K", 1536))
a = torch.randn(S, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(S, N, dtype=torch.bfloat16, device="cuda")
epi = os.environ.get("EPI", "gelu")
if epi == "resid":                      # the o-projection's epilogue: x += (a W^T + b) * gate, fp32 read-modify-write
    x = torch.randn(S, N, device="cuda")
    mod, e0 = torch.randn(6, N, device="cuda"), torch.randn(1, 6, N, device="cuda")
if epi == "qkv":                        # round 5: q | k | v in one launch, V^T stored transposed (OMH_EPI_BF16_SPLIT_T)
    d = K
    w3 = (torch.randn(3 * d, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    b3 = torch.randn(3 * d, device="cuda")
    qk = torch.empty(S, 2 * d, dtype=torch.bfloat16, device="cuda")
    Sp = (S + 63) // 64 * 64
    vt = torch.zeros(d, Sp, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    if epi == "qkv":
        ops.gemm_raw(ops.ptr(a), ops.ptr(w3), ops.ptr(qk), S, 3 * d, K, K, K, 2 * d, ops.EPI_BF16_SPLIT_T, bias=ops.ptr(b3),
                     bias_mode=ops.BIAS_N, aux=ops.ptr(vt), ldaux=Sp, n_split=2 * d)
    elif epi == "resid":
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), S, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias), bias_mode=ops.BIAS_N,
                     gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N, gate_rows=S, gate_const=0.0)
    else:
        ops.gemm(a, w, out=out, bias=bias, epilogue=ops.EPI_GELU_BF16 if epi == "gelu" else ops.EPI_BF16)
torch.cuda.synchronize()
