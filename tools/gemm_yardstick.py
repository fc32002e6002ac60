"""Yardstick for the DiT block GEMMs — TOOLS ONLY: never imported by the product, the tests or
bench.py.  On ONE box, interleaved, for the six GEMM shapes of a Wan2.1-1.3B block at S = 32 760:

  * this library's launch WITH its fused epilogue (bias / GELU / gated fp32 residual / transposed V), and
  * the vendor library behind torch (hipBLASLt / rocBLAS through ``torch.mm`` / ``torch.addmm`` on bf16 operands) for the
    BARE product (bf16 out, no epilogue — i.e. LESS work than the launch it is compared with: the elementwise passes a
    library-based block would still have to run are timed separately as ``+epilogue`` with plain torch ops).

    python tools/gemm_yardstick.py [S] > gpurun_out/r06_gemm_yardstick.json

The gap between the two columns is what is left to a better kernel on this part; frac = TFLOP/s / 2 500.
"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("omnihuman-1-hack_amd.ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32760
D, F = 1536, 8960
dev = "cuda"
torch.manual_seed(0)


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3            # us


def mk(M, N, K):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    return a, w, torch.randn(N, device=dev)


ptr = ops.ptr
rows = {}
mod = torch.randn(6, D, device=dev)
e0 = torch.randn(1, 6, D, device=dev)


def shape(name, M, N, K, ours, lib_bare, lib_epi):
    res = {"M_N_K": [M, N, K], "tflop": 2.0 * M * N * K / 1e12}
    t = {"ours_fused": [], "library_bare": [], "library_plus_epilogue": []}
    for _ in range(3):                                   # interleaved: clocks drift with temperature / power
        t["ours_fused"].append(timed(ours))
        t["library_bare"].append(timed(lib_bare))
        t["library_plus_epilogue"].append(timed(lib_epi))
    for k, v in t.items():
        us = sorted(v)[1]
        res[k] = {"us": round(us, 1), "tflops": round(res["tflop"] * 1e6 / us, 1),
                  "mfma_frac": round(res["tflop"] * 1e6 / us / 2500.0, 4), "runs_us": [round(x, 1) for x in v]}
    rows[name] = res
    sys.stderr.write("%s %s\n" % (name, json.dumps(res)))


# ---- q | k | v in one launch (V^T transposed) vs three library products' worth as one [S, 3d] product
a, w, b = mk(S, 3 * D, D)
qk = torch.empty(S, 2 * D, dtype=torch.bfloat16, device=dev)
Sp = (S + 63) // 64 * 64
vt = torch.zeros(D, Sp, dtype=torch.bfloat16, device=dev)
out3 = torch.empty(S, 3 * D, dtype=torch.bfloat16, device=dev)
bb = b.bfloat16()
shape("qkv_proj_fused", S, 3 * D, D,
      lambda: ops.gemm_raw(ptr(a), ptr(w), ptr(qk), S, 3 * D, D, D, D, 2 * D, ops.EPI_BF16_SPLIT_T, bias=ptr(b),
                           bias_mode=ops.BIAS_N, aux=ptr(vt), ldaux=Sp, n_split=2 * D),
      lambda: torch.mm(a, w.t(), out=out3),
      lambda: (torch.addmm(bb, a, w.t(), out=out3), out3[:, 2 * D:].t().contiguous()))

# ---- o-proj + gate + fp32 residual
a1, w1, b1 = mk(S, D, D)
x = torch.randn(S, D, device=dev)
ob = torch.empty(S, D, dtype=torch.bfloat16, device=dev)
gate = torch.randn(D, device=dev)
shape("o_proj_gate_resid", S, D, D,
      lambda: ops.gemm_raw(ptr(a1), ptr(w1), ptr(x), S, D, D, D, D, D, ops.EPI_RESID, bias=ptr(b1), bias_mode=ops.BIAS_N,
                           gate0=ptr(mod, 2 * D), gate1=ptr(e0, 2 * D), gate1_stride=6 * D, gate_rows=S, gate_const=0.0),
      lambda: torch.mm(a1, w1.t(), out=ob),
      lambda: (torch.mm(a1, w1.t(), out=ob), x.addcmul_(ob.float() + b1, gate)))

# ---- cross q (bf16 out + bias)
shape("cross_q_proj", S, D, D,
      lambda: ops.gemm_raw(ptr(a1), ptr(w1), ptr(ob), S, D, D, D, D, D, ops.EPI_BF16, bias=ptr(b1), bias_mode=ops.BIAS_N),
      lambda: torch.mm(a1, w1.t(), out=ob),
      lambda: torch.addmm(b1.bfloat16(), a1, w1.t(), out=ob))

# ---- FFN up + GELU-tanh
a2, w2, b2 = mk(S, F, D)
hb = torch.empty(S, F, dtype=torch.bfloat16, device=dev)
b2b = b2.bfloat16()
shape("ffn_up_gemm_gelu", S, F, D,
      lambda: ops.gemm_raw(ptr(a2), ptr(w2), ptr(hb), S, F, D, D, D, F, ops.EPI_GELU_BF16, bias=ptr(b2), bias_mode=ops.BIAS_N),
      lambda: torch.mm(a2, w2.t(), out=hb),
      lambda: (torch.addmm(b2b, a2, w2.t(), out=hb), torch.nn.functional.gelu(hb, approximate="tanh")))

# ---- FFN down + gate + residual
a3, w3, b3 = mk(S, D, F)
shape("ffn_down_gate_resid", S, D, F,
      lambda: ops.gemm_raw(ptr(a3), ptr(w3), ptr(x), S, D, F, F, F, D, ops.EPI_RESID, bias=ptr(b3), bias_mode=ops.BIAS_N,
                           gate0=ptr(mod, 5 * D), gate1=ptr(e0, 5 * D), gate1_stride=6 * D, gate_rows=S, gate_const=0.0),
      lambda: torch.mm(a3, w3.t(), out=ob),
      lambda: (torch.mm(a3, w3.t(), out=ob), x.addcmul_(ob.float() + b3, gate)))

agg = {}
for col in ("ours_fused", "library_bare", "library_plus_epilogue"):
    names = ("qkv_proj_fused", "o_proj_gate_resid", "ffn_up_gemm_gelu", "ffn_down_gate_resid")
    tf = sum(rows[n]["tflop"] for n in names)
    us = sum(rows[n][col]["us"] for n in names)
    agg[col] = {"us_per_block": round(us, 1), "aggregate_frac": round(tf * 1e6 / us / 2500.0, 4)}
print(json.dumps({"S": S, "device": torch.cuda.get_device_name(0), "torch": torch.__version__,
                  "blas": "torch.mm / addmm on bf16 (hipBLASLt / rocBLAS as torch selects)", "shapes": rows,
                  "block_aggregate_qkv_o_ffnup_ffndown": agg,
                  "note": "library_bare is the product alone (bf16 out, no bias / GELU / residual / transpose): less work "
                          "than ours_fused; library_plus_epilogue adds the elementwise passes as plain torch ops"}, indent=1))
