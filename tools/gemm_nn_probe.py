"""k-major-B (dgrad on W as stored) vs row-major-B GEMM on the training step's dgrad shapes: us per call."""
import importlib, json, math, os, sys
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

res = {}
for M, N, K in [(6240, 1536, 1536), (6240, 1536, 3072), (6240, 1536, 8960), (6240, 8960, 1536), (2048, 1536, 1536),
                (1560, 1536, 8960), (32760, 1536, 1536)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(K, N, device="cuda") / math.sqrt(K)).bfloat16()      # [K, N]: k-major B
    wt = w.t().contiguous()
    out = torch.empty(M, N, device="cuda")
    r = {}
    for tile in ("big", "small"):
        os.environ["OMH_GEMM_TILE"] = tile
        r[tile + "_kmajor"] = round(t(lambda: ops.gemm(a, w, out=out, epilogue=ops.EPI_F32, b_kmajor=True)), 1)
        r[tile + "_rowmajor"] = round(t(lambda: ops.gemm(a, wt, out=out, epilogue=ops.EPI_F32)), 1)
    del os.environ["OMH_GEMM_TILE"]
    r["auto_kmajor"] = round(t(lambda: ops.gemm(a, w, out=out, epilogue=ops.EPI_F32, b_kmajor=True)), 1)
    r["auto_rowmajor"] = round(t(lambda: ops.gemm(a, wt, out=out, epilogue=ops.EPI_F32)), 1)
    r["transpose"] = round(t(lambda: ops.transpose_bf16(w)), 1)
    res[f"{M}x{N}x{K}"] = r
    print(f"{M}x{N}x{K}", r, flush=True)
