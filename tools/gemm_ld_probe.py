"""L2 channel-camping probe: the S=32760 GEMMs with the operands' row strides padded by `pad` elements.
A bf16 row of K=1536 is 3072 B = 12 x 256 B: if the L2 interleaves its 16 channels at 256 B, the 256 rows of a
k-slice (128 B each, 3072 B apart) fall on only gcd-limited channels.  GPU box only."""
import importlib, json, math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S = 32760
res = {}


def run(N, K, pad_a, pad_b, epi, pad_c=0, reps=20):
    lda, ldb = K + pad_a, K + pad_b
    a = torch.randn(S, lda, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, ldb, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    ldc = N + pad_c
    odt = torch.float32 if epi in (ops.EPI_F32, ops.EPI_RESID) else torch.bfloat16
    out = torch.zeros(S, ldc, dtype=odt, device="cuda")
    gate = torch.randn(1, N, device="cuda")

    def go():
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), S, N, K, lda, ldb, ldc, epi, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate1=ops.ptr(gate) if epi == ops.EPI_RESID else None, gate1_stride=N,
                     gate_rows=S)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        go()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return round(ms * 1e3, 1), round(2.0 * S * N * K / ms / 1e9, 1)


for name, N, K, epi in (("ffn1_gelu", 8960, 1536, ops.EPI_GELU_BF16), ("ffn1_bias", 8960, 1536, ops.EPI_BF16),
                        ("ffn2_resid", 1536, 8960, ops.EPI_RESID), ("qk", 3072, 1536, ops.EPI_BF16),
                        ("oproj_resid", 1536, 1536, ops.EPI_RESID)):
    for pad in (0, 8, 32, 64, 128):
        res[f"{name}[pad={pad}]"] = run(N, K, pad, pad, epi)
    res[f"{name}[padA=64 only]"] = run(N, K, 64, 0, epi)
    res[f"{name}[padB=64 only]"] = run(N, K, 0, 64, epi)
    if epi != ops.EPI_RESID:
        res[f"{name}[pad=64,padC=64]"] = run(N, K, 64, 64, epi, pad_c=64)
print(json.dumps(res, indent=1))
