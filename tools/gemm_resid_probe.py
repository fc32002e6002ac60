"""The gated-residual o-projection at S = 32 760 (x += (o Wo^T + b) * gate, K = 1536) on the three streams that take it —
256 x 384 (old C in the epilogue), 256 x 192 (old C in the k loop), 256 x 256 (round 5: 10 of 16 tiles in the k loop) —
interleaved on one box; also the training step's M = 6 240 and the 14B width."""
import importlib, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
g = torch.Generator(device="cuda").manual_seed(1)


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = {}
for M, N, K in ((32760, 1536, 1536), (21840, 1536, 1536), (6240, 1536, 1536), (32760, 5120, 5120)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    mod = torch.randn(6, N, device="cuda", generator=g)
    e0 = torch.randn(1, 6, N, device="cuda", generator=g)
    x = torch.randn(M, N, device="cuda", generator=g)

    def run():
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias), bias_mode=ops.BIAS_N,
                     gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N, gate_rows=M, gate_const=0.0)
    res = {"r384_us": [], "r192_us": [], "r256_us": []}
    for _ in range(3):
        for name, opts in (("r384_us", dict(GEMM_KERNEL="w64", GEMM_W64_R192="0", GEMM_W64_R256="0")),
                           ("r192_us", dict(GEMM_KERNEL="w64", GEMM_W64_R192="1", GEMM_W64_R256="0")),
                           ("r256_us", dict(GEMM_KERNEL="w64", GEMM_W64_R192=None, GEMM_W64_R256="1"))):
            with ops.options(**opts):
                res[name].append(round(t(run), 1))
    fl = 2.0 * M * N * K
    res["best_tflops"] = {k_: round(fl / min(v) / 1e6, 1) for k_, v in res.items()}
    out[f"{M}x{N}x{K}"] = res
    del a, w, x
print(json.dumps(out))
