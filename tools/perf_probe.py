"""Times the hot kernels at BASELINE config-2 shapes (S=32760, Wan2.1-1.3B) with HIP events."""
import importlib
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    S = int(os.environ.get("S", 32760))
    d, f, H, D = 1536, 8960, 12, 128
    res = {}
    x = torch.randn(S, d, device="cuda").to(torch.bfloat16)
    for name, N, K, epi in (("qk_f32", 2 * d, d, ops.EPI_F32), ("o_bf16", d, d, ops.EPI_BF16),
                            ("ffn1_gelu", f, d, ops.EPI_GELU_BF16), ("ffn2_bf16", d, f, ops.EPI_BF16)):
        a = torch.randn(S, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(S, N, dtype=torch.float32 if epi == ops.EPI_F32 else torch.bfloat16, device="cuda")
        ms = timeit(lambda: ops.gemm(a, w, out=out, bias=bias, epilogue=epi))
        res[name] = {"ms": ms, "tflops": 2.0 * S * N * K / ms / 1e9}
    # gated-residual epilogue (o-projection): x += (a w^T + b) * gate
    a = torch.randn(S, d, device="cuda").to(torch.bfloat16)
    w = (torch.randn(d, d, device="cuda") / math.sqrt(d)).to(torch.bfloat16)
    bias = torch.randn(d, device="cuda")
    xres = torch.randn(S, d, device="cuda")
    g0, g1 = torch.randn(d, device="cuda"), torch.randn(1, d, device="cuda")
    ms = timeit(lambda: ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(xres), S, d, d, d, d, d, ops.EPI_RESID,
                                     bias=ops.ptr(bias), bias_mode=ops.BIAS_N, gate0=ops.ptr(g0), gate1=ops.ptr(g1),
                                     gate1_stride=d, gate_rows=S, gate_const=0.0))
    res["o_resid"] = {"ms": ms, "tflops": 2.0 * S * d * d / ms / 1e9}
    q = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
    Sp = (S + 63) // 64 * 64
    vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :S] = torch.randn(1, H * D, S, device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    runs = sorted(timeit(lambda: ops.flash_attn(q, k, vt, None, out=o), iters=4, warm=1) for _ in range(5))
    ms = runs[len(runs) // 2]
    res["self_attn"] = {"ms": ms, "ms_min": runs[0], "ms_max": runs[-1], "tflops": 4.0 * S * S * H * D / ms / 1e9}
    xf = torch.randn(S, d, device="cuda")
    mod = torch.randn(6, d, device="cuda")
    y = torch.empty(S, d, dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: ops.layernorm_modulate_raw(ops.ptr(xf), ops.ptr(y), S, d, 1e-6, 1.0, ops.ptr(mod, d), None, 0,
                                                   ops.ptr(mod), None, 0, S))
    res["ln_mod"] = {"ms": ms, "GBps": S * d * 6 / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
