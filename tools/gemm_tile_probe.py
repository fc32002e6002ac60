"""Tile configurations of omh_gemm_bf16 on the shapes of the training step and the single-frame forward (GPU box):
    python tools/gemm_tile_probe.py
For each shape: us per call with the dispatcher's own choice and with every forced configuration."""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


shapes = [(6240, 1536, 1536, ops.EPI_BF16), (6240, 1536, 1536, ops.EPI_RESID), (6240, 1536, 4608, ops.EPI_F32),
          (6240, 1536, 8960, ops.EPI_RESID), (6240, 3072, 1536, ops.EPI_BF16), (6240, 8960, 1536, ops.EPI_GELU_BF16),
          (6240, 8960, 1536, ops.EPI_BF16), (1560, 1536, 1536, ops.EPI_BF16), (1560, 1536, 8960, ops.EPI_RESID),
          (3120, 1536, 1536, ops.EPI_BF16), (3120, 8960, 1536, ops.EPI_GELU_BF16), (3120, 1536, 8960, ops.EPI_RESID),
          (24960, 1536, 1536, ops.EPI_BF16), (24960, 1536, 8960, ops.EPI_RESID), (32760, 1536, 1536, ops.EPI_RESID)]
res = []
for M, N, K, epi in shapes:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi in (ops.EPI_F32, ops.EPI_RESID) else torch.bfloat16)
    kw = dict(gate_const=1.0) if epi == ops.EPI_RESID else {}
    fn = lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, epi, bias=ops.ptr(b), bias_mode=ops.BIAS_N, **kw)
    row = {"M": M, "N": N, "K": K, "epi": epi}
    ops.set_option("OMH_GEMM_TILE", None)
    ops.set_option("OMH_GEMM_KERNEL", None)
    row["auto_us"] = round(timeit(fn), 1)
    for t in ("big", "mid192", "small", "tiny"):
        ops.set_option("OMH_GEMM_TILE", t)
        row[t + "_us"] = round(timeit(fn), 1)
    ops.set_option("OMH_GEMM_TILE", None)
    ops.set_option("OMH_GEMM_KERNEL", "w64")
    try:
        row["w64_us"] = round(timeit(fn), 1)
    except Exception as e:
        row["w64_us"] = None
    ops.set_option("OMH_GEMM_KERNEL", None)
    row["auto_tflops"] = round(2 * M * N * K / row["auto_us"] / 1e6, 0)
    res.append(row)
    print(json.dumps(row), flush=True)
