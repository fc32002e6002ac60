"""Micro-benchmark of one GEMM call under environment switches, interleaved on one box:
    python tools/gemm_kind_probe.py EPI M N K VAR=val [VAR=val ...]      (EPI: f32 | bf16 | gelu | gelubwd | resid)
prints us / TFLOP/s with the variables unset and set, alternating twice."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
epi_name, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
env = dict(a.split("=", 1) for a in sys.argv[5:])
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda"); pre = torch.randn(M, N, device="cuda").bfloat16()
x = torch.randn(M, N, device="cuda"); outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
mod = torch.randn(6, N, device="cuda"); e0 = torch.randn(2, 6, N, device="cuda")
def run():
    if epi_name == "resid":
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias), bias_mode=ops.BIAS_N,
                     gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N, gate_rows=(M + 1) // 2, gate_const=0.0)
    elif epi_name == "gelubwd":
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(outb), M, N, K, K, K, N, ops.EPI_GELU_BWD_BF16, aux=ops.ptr(pre), ldaux=N)
    elif epi_name == "f32":
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_F32, bias=ops.ptr(bias), bias_mode=ops.BIAS_N)
    else:
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(outb), M, N, K, K, K, N, ops.EPI_GELU_BF16 if epi_name == "gelu" else ops.EPI_BF16,
                     bias=ops.ptr(bias), bias_mode=ops.BIAS_N)
def timed():
    for _ in range(5): run()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / 50 * 1e3
res = []
for rep in range(2):
    for on in (False, True):
        for k, v in env.items():
            if on: os.environ[k] = v
            else: os.environ.pop(k, None)
        us = timed(); res.append("%s %.1fus/%.0fTF" % ("set" if on else "unset", us, 2 * M * N * K / us / 1e6))
for k in env: os.environ.pop(k, None)
print(epi_name, M, N, K, env, " | ".join(res))
