"""Small-S GEMM timings (training clip / config-1 shapes, S=1560)."""
import importlib, math, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

res = {}
for (M, N, K, epi, name) in ((1560, 1536, 1536, ops.EPI_F32, "proj_f32"), (1560, 3072, 1536, ops.EPI_BF16, "qk_bf16"),
                             (1560, 8960, 1536, ops.EPI_GELU_BF16, "ffn1"), (1560, 1536, 8960, ops.EPI_F32, "ffn2"),
                             (1536, 1536, 1560, ops.EPI_F32, "wgrad"), (512, 1536, 1536, ops.EPI_F32, "ctx_kv"),
                             (1536, 8960, 1560, ops.EPI_F32, "wgrad_ffn")):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if epi == ops.EPI_F32 else torch.bfloat16, device="cuda")
    us = timeit(lambda: ops.gemm(a, w, out=out, epilogue=epi))
    res[name] = {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
print(json.dumps(res))
