"""The training step's self-attention FORWARD (B clips of S = 1560, lse + fp32 output requested): the short-sequence kernel the
step pins (OMH_ATTN_SHORT_KERNEL) against what the dispatcher would pick without the pin (the long-sequence stream from its
size threshold on).  us per call."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
H, D, S = 12, 128, int(os.environ.get("S", 1560))
d = H * D
for B in (1, 4, 16):
    q, k = [torch.randn(B * S, d, device="cuda").bfloat16() for _ in range(2)]
    Sp = (S + 63) // 64 * 64
    vt = torch.randn(B, d, Sp, device="cuda").bfloat16()
    o = torch.empty(B * S, d, device="cuda", dtype=torch.bfloat16); o32 = torch.empty(B * S, d, device="cuda"); lse = torch.empty(B, H, S, device="cuda")
    for name, flags, kern in (("short kernel (pinned)", ops.ATTN_SHORT_KERNEL, None), ("short + split", ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT, None),
                              ("dispatcher's choice", 0, None), ("forced w64", 0, "w64")):
        ops.set_option("OMH_ATTN_KERNEL", kern)
        f = lambda: ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(o), None, B, H, S, S, S * d, d, S * d, d, d * Sp, S * d, d, Sp,
                                       D ** -0.5, lse=ops.ptr(lse), q_prescaled=1, o32=ops.ptr(o32), flags=flags)
        try:
            for _ in range(3): f()
        except Exception as ex:
            print(B, name, "not available:", str(ex)[:80]); continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        print(f"B={B:2d} {name:24s} {us:8.1f} us  {4.0 * S * S * d * B / us / 1e6:7.0f} TFLOP/s", flush=True)
    ops.set_option("OMH_ATTN_KERNEL", None)
