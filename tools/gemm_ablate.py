"""GEMM ablations: per-k-step cost and fixed per-tile cost from a K sweep, and the
same with every row aliased to row 0 (lda=ldb=0: all operand traffic L2/TCP resident)."""
import importlib, math, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S = 32760

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

res = {}
for N in (1536, 8960):
    for K in (512, 1536, 3072, 6144):
        a = torch.randn(S, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
        out = torch.empty(S, N, dtype=torch.bfloat16, device="cuda")
        for name, lda, ldb in (("real", K, K), ("alias", 0, 0)):
            ms = timeit(lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), S, N, K, lda, ldb, N, ops.EPI_BF16))
            res[f"N{N}_K{K}_{name}"] = {"ms": round(ms, 4), "tflops": round(2.0 * S * N * K / ms / 1e9, 1)}
print(json.dumps(res, indent=1))
