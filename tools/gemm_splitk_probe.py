"""Split K (ABI v9) against the unsplit dispatch on the few-row long-contraction products of the DiT (FFN-down with the
gated residual, FFN-up input gradient): us per product with OMH_GEMM_SPLITK unset / 0, interleaved."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def t(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for M, N, K in ((1560, 1536, 8960), (3120, 1536, 8960), (1560, 1536, 4608), (3120, 1536, 4608), (6240, 1536, 8960)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    x = torch.zeros(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    gate = torch.randn(N, device="cuda")
    resid = lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                                 bias_mode=ops.BIAS_N, gate0=ops.ptr(gate), gate_const=0.0, split_k=True)
    f32 = lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_F32, split_k=True)
    row = {}
    for rnd in range(2):
        for mode in ("split", "unsplit"):
            if mode == "unsplit":
                ops.set_option("OMH_GEMM_SPLITK", "0")
            else:
                ops.set_option("OMH_GEMM_SPLITK", None)
            row.setdefault(mode + " resid", []).append(round(t(resid), 1))
            row.setdefault(mode + " f32", []).append(round(t(f32), 1))
    ops.set_option("OMH_GEMM_SPLITK", None)
    print((M, N, K), row, "TFLOP/s split resid", round(2.0 * M * N * K / min(row["split resid"]) / 1e6, 1), flush=True)
