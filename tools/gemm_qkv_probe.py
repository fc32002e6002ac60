"""The self-attention projections of one block at S = 32 760: q|k (N = 3072) + V^T as two launches against the fused
q|k|v launch (OMH_EPI_BF16_SPLIT_T), interleaved on one box; also the gated-residual o-projection for reference."""
import importlib, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
M, d = 32760, 1536
Sp = (M + 63) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(1)
h = torch.randn(M, d, device="cuda", generator=g).bfloat16()
w = (torch.randn(3 * d, d, device="cuda", generator=g) / math.sqrt(d)).bfloat16()
bias = torch.randn(3 * d, device="cuda", generator=g)
qk = torch.empty(M, 2 * d, dtype=torch.bfloat16, device="cuda")
vt = torch.zeros(d, Sp, dtype=torch.bfloat16, device="cuda")


def fused():
    ops.gemm_raw(ops.ptr(h), ops.ptr(w), ops.ptr(qk), M, 3 * d, d, d, d, 2 * d, ops.EPI_BF16_SPLIT_T, bias=ops.ptr(bias),
                 bias_mode=ops.BIAS_N, aux=ops.ptr(vt), ldaux=Sp, n_split=2 * d)


def separate():
    ops.gemm_raw(ops.ptr(h), ops.ptr(w), ops.ptr(qk), M, 2 * d, d, d, d, 2 * d, ops.EPI_BF16, bias=ops.ptr(bias), bias_mode=ops.BIAS_N)
    ops.gemm_raw(ops.ptr(w, 2 * d * d), ops.ptr(h), ops.ptr(vt), d, M, d, d, d, Sp, ops.EPI_BF16, bias=ops.ptr(bias, 2 * d),
                 bias_mode=ops.BIAS_M)


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


res = {"fused_us": [], "separate_us": []}
for _ in range(4):
    res["fused_us"].append(round(t(fused), 1))
    res["separate_us"].append(round(t(separate), 1))
fl = 2.0 * M * 3 * d * d
res["fused_tflops"] = round(fl / min(res["fused_us"]) / 1e6, 1)
res["separate_tflops"] = round(fl / min(res["separate_us"]) / 1e6, 1)
print(json.dumps(res))

# V^T row pitch: roundup(S, 64) = 32 768 columns = 65 536 bytes, a power of two — every one of the 32 rows a store (and
# an attention V^T tile read) touches then maps to the same HBM channel group.  The same launch with 64 more columns:
Sp2 = Sp + 64
vt2 = torch.zeros(d, Sp2, dtype=torch.bfloat16, device="cuda")


def fused_padded():
    ops.gemm_raw(ops.ptr(h), ops.ptr(w), ops.ptr(qk), M, 3 * d, d, d, d, 2 * d, ops.EPI_BF16_SPLIT_T, bias=ops.ptr(bias),
                 bias_mode=ops.BIAS_N, aux=ops.ptr(vt2), ldaux=Sp2, n_split=2 * d)


q = torch.randn(1, M, 12, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(1, M, 12, 128, device="cuda", generator=g).bfloat16()
vt.normal_(generator=g)
vt[:, M:] = 0
vt2[:, :M] = vt[:, :M]
o = torch.empty(1, M, 12, 128, dtype=torch.bfloat16, device="cuda")
res2 = {"fused_pitch_65536B_us": [], "fused_pitch_65664B_us": [], "attn_pitch_65536B_ms": [], "attn_pitch_65664B_ms": []}
for _ in range(3):
    res2["fused_pitch_65536B_us"].append(round(t(fused), 1))
    res2["fused_pitch_65664B_us"].append(round(t(fused_padded), 1))
    res2["attn_pitch_65536B_ms"].append(round(t(lambda: ops.flash_attn(q, k, vt.view(1, d, Sp), out=o), 6) / 1e3, 4))
    res2["attn_pitch_65664B_ms"].append(round(t(lambda: ops.flash_attn(q, k, vt2.view(1, d, Sp2), out=o), 6) / 1e3, 4))
print(json.dumps(res2))
