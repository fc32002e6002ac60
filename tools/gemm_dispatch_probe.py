"""Does omh_gemm_bf16's default choice between the 8-wave kernels and the 256x384 stream kernel pick the faster one?
Times OMH_GEMM_KERNEL = 8w / w64 / unset on the shapes of the training step, config 4 and the teacher pair."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def t(a, w, bias, epi, kernel):
    if kernel:
        os.environ["OMH_GEMM_KERNEL"] = kernel
    else:
        os.environ.pop("OMH_GEMM_KERNEL", None)
    for _ in range(3):
        ops.gemm(a, w, bias=bias, epilogue=epi)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.gemm(a, w, bias=bias, epilogue=epi)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3


for M, N, K in ((6240, 3072, 1536), (6240, 8960, 1536), (6240, 1536, 1536), (6240, 1536, 8960), (3120, 3072, 1536),
                (3120, 8960, 1536), (1560, 8960, 1536), (21840, 1536, 1536), (21840, 3072, 1536), (21840, 8960, 1536),
                (21840, 1536, 8960), (12480, 1536, 1536), (12480, 3072, 1536), (32760, 5120, 5120), (32760, 13824, 5120)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    r = {k: min(t(a, w, bias, ops.EPI_BF16, k) for _ in range(2)) for k in ("8w", "w64", "")}
    best = min(r["8w"], r["w64"])
    print(f"M{M} N{N} K{K}: 8w {r['8w']:.1f} us  w64 {r['w64']:.1f} us  default {r['']:.1f} us  {'OK' if r[''] <= best * 1.04 else 'MISS'}", flush=True)
