"""Cross-attention shape of the DiT block (Lq = 32760 video tokens, Lk = 512 padded text tokens, 120 / 40 of them real):
the short-sequence kernel against the long-sequence stream kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
B, H, Lq, Lk, D = 1, 12, 32760, 512, 128
q = torch.randn(B, Lq, H, D, device="cuda").bfloat16()
k = torch.randn(B, Lk, H, D, device="cuda").bfloat16()
v = torch.randn(B, Lk, H, D, device="cuda").bfloat16()
vt = torch.zeros(B, H * D, Lk, dtype=torch.bfloat16, device="cuda")
vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
for real in (120, 40, 512):
    kl = torch.tensor([real], dtype=torch.int32, device="cuda")
    outs = {}
    for kern in ("base", "w64"):
        os.environ["OMH_ATTN_KERNEL"] = kern
        f = lambda: ops.flash_attn(q, k, vt, kl)
        o = f(); f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        outs[kern] = o
        print(f"k_len {real} {kern}: {s.elapsed_time(e) / 10 * 1e3:.1f} us", flush=True)
    d = (outs["base"].float() - outs["w64"].float())
    print(f"   rel diff base vs w64 {float(d.norm() / outs['base'].float().norm()):.2e}", flush=True)
