"""The 256 x 384 one-wave-per-SIMD GEMM stream (csrc/gemm_w64.hip) against the 8-wave kernels of gemm_bf16.hip: every
epilogue bit for bit on the whole output (same accumulation order over k, same epilogue arithmetic), fp32 parity on
sampled rows, and interleaved timing on the DiT's large shapes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
P = ops.ptr


def run(kernel, epi, a, w, bias=None, out=None, gate0=None, gate1=None, gate_rows=1, gate_const=0.0):
    ops.set_option("OMH_GEMM_KERNEL", kernel)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if epi in (ops.EPI_BF16, ops.EPI_GELU_BF16) else torch.float32)
    ops.gemm_raw(P(a), P(w), P(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), epi,
                 bias=P(bias) if bias is not None else None, bias_mode=ops.BIAS_N if bias is not None else ops.BIAS_NONE,
                 gate0=P(gate0) if gate0 is not None else None, gate1=P(gate1) if gate1 is not None else None,
                 gate1_stride=gate1.stride(0) if gate1 is not None else 0, gate_rows=gate_rows, gate_const=gate_const)
    return out


def check(M, N, K, gate_rows=None):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    ok = True
    for name, epi, b in (("f32", ops.EPI_F32, None), ("f32+b", ops.EPI_F32, bias), ("bf16", ops.EPI_BF16, None),
                         ("bf16+b", ops.EPI_BF16, bias), ("gelu+b", ops.EPI_GELU_BF16, bias)):
        got, old = run("w64", epi, a, w, b), run("8w", epi, a, w, b)
        same = torch.equal(got.view(torch.int16 if got.dtype == torch.bfloat16 else torch.int32),
                           old.view(torch.int16 if got.dtype == torch.bfloat16 else torch.int32))
        d = float((got.float() - old.float()).abs().max())
        print(f"check M{M} N{N} K{K} {name}: bit-identical {same} (max diff {d:.3e})", flush=True)
        ok &= same
    gr = gate_rows or M
    nb = (M + gr - 1) // gr
    g0 = torch.randn(N, device="cuda", generator=g)
    g1 = torch.randn(nb, 6, N, device="cuda", generator=g)[:, 2]              # strided rows, as the modulation table
    c0 = torch.randn(M, N, device="cuda", generator=g)
    for name, kw in (("resid g0+g1+b", dict(bias=bias, gate0=g0, gate1=g1, gate_rows=gr)),
                     ("resid const", dict(gate_const=1.0)), ("resid g1", dict(gate1=g1, gate_rows=gr, bias=bias))):
        got = run("w64", ops.EPI_RESID, a, w, out=c0.clone(), **kw)
        old = run("8w", ops.EPI_RESID, a, w, out=c0.clone(), **kw)
        same = torch.equal(got.view(torch.int32), old.view(torch.int32))
        print(f"check M{M} N{N} K{K} {name} rows/gate {gr}: bit-identical {same} (max diff {float((got - old).abs().max()):.3e})", flush=True)
        ok &= same
    rows = torch.tensor([0, 1, 31, 32, 127, 128, 255, 256, M - 257, M - 2, M - 1], device="cuda")
    got = run("w64", ops.EPI_F32, a, w, bias)
    ref = a[rows].float() @ w.float().t() + bias
    e = float((got[rows] - ref).norm() / ref.norm())
    print(f"check M{M} N{N} K{K}: rel {e:.3e} vs fp32 torch on sampled rows", flush=True)
    return ok and e < 2e-5


def timeit(M, N, K, epi, name):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    kw = dict(bias=bias)
    out = None
    if epi == ops.EPI_RESID:
        out = torch.randn(M, N, device="cuda")
        kw.update(gate0=torch.randn(N, device="cuda"), gate1=torch.randn(2, N, device="cuda"), gate_rows=(M + 1) // 2)
    res = {}
    for rnd in range(3):
        for kernel in ("8w", "w64"):
            for _ in range(2):
                run(kernel, epi, a, w, out=out, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run(kernel, epi, a, w, out=out, **kw)
            e.record(); torch.cuda.synchronize()
            res.setdefault(kernel, []).append(s.elapsed_time(e) / 10)
    for kernel in ("8w", "w64"):
        ms = sorted(res[kernel])[1]
        print(f"time M{M} N{N} K{K} {name} {kernel}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.0f} TF  {['%.1f' % (x * 1e3) for x in res[kernel]]}", flush=True)


if __name__ == "__main__":
    ok = True
    for shp in () if os.environ.get("TIME_ONLY") else ((512, 384, 256, None), (1000, 776, 320, 300), (2000, 1536, 448, 500), (32760, 8960, 1536, 16380), (32760, 1536, 8960, 16380)):
        ok &= check(*shp)
    print("CHECK", "PASS" if ok else "FAIL", flush=True)
    for M in (32760,):
        timeit(M, 1536, 1536, ops.EPI_BF16, "bf16+bias")
        timeit(M, 1536, 1536, ops.EPI_RESID, "resid")
        timeit(M, 8960, 1536, ops.EPI_GELU_BF16, "gelu")
        timeit(M, 1536, 8960, ops.EPI_RESID, "resid")
        timeit(M, 4608, 1536, ops.EPI_BF16, "bf16+bias")
