"""Timing of the w64 stream variants / ablations at the benchmark shape, interleaved rounds on one box."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
D, S, H = 128, int(os.environ.get("S", 32760)), 12
q = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
k = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
Sp = (S + 63) // 64 * 64
vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
vt[:, :, :S] = torch.randn(1, H * D, S, device="cuda").to(torch.bfloat16)
o = torch.empty_like(q)
variants = sys.argv[1:] or ["2"]      # "0" / "1": ablation builds of the generator only (OMH_ATTN_ABL, tools/attn_abl.sh)
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        if v == "base":
            ops.set_option("OMH_ATTN_KERNEL", "base")
        else:
            ops.set_option("OMH_ATTN_KERNEL", "w64"); ops.set_option("OMH_W64_VARIANT", v)
        for _ in range(2):
            ops.flash_attn(q, k, vt, None, out=o)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8):
            ops.flash_attn(q, k, vt, None, out=o)
        e.record(); torch.cuda.synchronize()
        res[v].append(s.elapsed_time(e) / 8)
for v in variants:
    ms = sorted(res[v])[1]
    print(f"variant {v}: median {ms:.4f} ms  {4.0 * S * S * H * D / ms / 1e9:.1f} TF   all {['%.3f' % x for x in res[v]]}")
