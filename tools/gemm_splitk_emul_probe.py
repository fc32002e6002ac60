"""Upper bound for an ordered split-K of the long-K residual GEMM at small M: the same work issued as `splits`
independent k-slices (batch dimension, separate fp32 outputs) against the single-launch kernel."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n

N, K = 1536, 8960
for M in (1560, 3120, 6240):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    out = torch.zeros(M, N, device="cuda")
    r = {}
    for tile in ("tiny", "small", "big"):
        os.environ["OMH_GEMM_TILE"] = tile
        r[tile] = round(t(lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, ops.EPI_RESID, gate_const=0.5)), 1)
    for splits in (2, 3, 4, 5):
        ks = (K // splits) // 64 * 64
        part = torch.empty(splits, M, N, device="cuda")
        for tile in ("small", "big"):
            os.environ["OMH_GEMM_TILE"] = tile
            r[f"{tile}x{splits}"] = round(t(lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(part), M, N, ks, K, K, N, ops.EPI_F32,
                                                                 batch=splits, strideA=ks, strideB=ks, strideC=M * N)), 1)
    del os.environ["OMH_GEMM_TILE"]
    print(M, r, flush=True)
