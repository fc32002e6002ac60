"""256x256 tile as 8 waves x (128x64) ["big"] vs 4 waves x (128x128), accumulators in AGPRs ["huge"]: us and TFLOP/s."""
import importlib, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n

for M, N, K, epi in [(32760, 1536, 1536, "EPI_BF16"), (32760, 3072, 1536, "EPI_BF16"), (32760, 8960, 1536, "EPI_GELU_BF16"),
                     (32760, 1536, 8960, "EPI_RESID"), (6240, 1536, 8960, "EPI_F32"), (6240, 8960, 1536, "EPI_BF16"),
                     (65520, 8960, 1536, "EPI_GELU_BF16"), (32760, 5120, 5120, "EPI_BF16")]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    e = getattr(ops, epi)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if "BF16" in epi else torch.float32)
    r = {}
    for tile in ("big", "huge"):
        ops.set_option("OMH_GEMM_TILE", tile)
        if epi == "EPI_RESID":
            f = lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, e, gate_const=0.5)
        else:
            f = lambda: ops.gemm(a, w, out=out, epilogue=e)
        us = t(f)
        r[tile] = (round(us, 1), round(2.0 * M * N * K / us / 1e6, 1))
    print(f"{M}x{N}x{K} {epi}", r, flush=True)
