"""Timing-only ablations of the p256 k loop (build with OMH_GW64_ABLATIONS=1 python gen_gemm_w64.py > gemm_w64_asm.inc):
fp32-output product 32760 x 1536 x K on GEMM_W64_P256 = 0 (shipped) / 1 / a..e.   python tools/gemm_p256_abl.py [K]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
M, N, K = 32760, 1536, int(sys.argv[1]) if len(sys.argv) > 1 else 8960
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
if os.environ.get("ZERO") == "1":            # constant operands: the multipliers draw little power, the clock stays up
    a.zero_(); w.zero_()
x = torch.empty(M, N, device="cuda"); ptr = ops.ptr
def run(): ops.gemm_raw(ptr(a), ptr(w), ptr(x), M, N, K, K, K, N, ops.EPI_F32)
if os.environ.get("SET") == "m16":
    names = {"0": "shipped 256x384 stream", "m": "the same on 16x16x32 MFMAs"}
elif os.environ.get("SET") == "ldmod":
    names = {"0": "shipped dispatch", "1": "p256", "a": "loads nt", "b": "loads sc1", "c": "loads sc0", "d": "loads sc0 sc1", "e": "loads sc1 nt"}
else:
    names = {"0": "shipped dispatch", "1": "p256", "a": "no global loads", "b": "no LDS writes", "c": "no loads, no writes", "d": "no barrier", "e": "no fragment reads"}
for rep in range(2):
    for v in ("0m" if os.environ.get("SET") == "m16" else "01abcde"):
        ops.set_option("GEMM_W64_P256", v)
        for _ in range(3): run()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) / 10 * 1e3
        print(f"K={K} {v} {names[v]:22s} {us:8.1f} us  frac {2.0 * M * N * K / us / 1e6 / 2500:.3f}", flush=True)
