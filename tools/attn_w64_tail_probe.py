"""Split-KV tail of the 4x64 attention kernel at BASELINE config 4's sequence length (S = 21 840, 12 heads: 1 032
tiles = 4 x 256 + 8): timing with / without the split (OMH_W64_SPLIT=0) and sampled-row parity of the tail rows."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
D, S, H = 128, int(os.environ.get("S", 21840)), 12
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(1, S, H, D, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(1, S, H, D, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(1, S, H, D, device="cuda", generator=g).to(torch.bfloat16)
Sp = (S + 63) // 64 * 64
vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
vt[:, :, :S] = v.reshape(1, S, H * D).transpose(1, 2)
outs = {}
for split in ("1", "0", "1", "0"):
    os.environ["OMH_W64_SPLIT"] = split
    o = torch.empty_like(q)
    for _ in range(3):
        ops.flash_attn(q, k, vt, None, out=o)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.flash_attn(q, k, vt, None, out=o)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    outs[split] = o.clone()
    print(f"split={split}: {ms:.4f} ms  {4.0 * S * S * H * D / ms / 1e9:.1f} TF", flush=True)
rows = torch.tensor([0, 255, 19967, 19968, 20000, 21000, S - 257, S - 2, S - 1], device="cuda")
for h in (0, 11):
    s_ = (q[0, rows, h].float() @ k[0, :, h].float().t()) * D ** -0.5
    ref = torch.softmax(s_, -1) @ v[0, :, h].float()
    for sp in ("1", "0"):
        got = outs[sp][0, rows, h].float()
        print(f"head {h} split={sp}: rel {float((got - ref).norm() / ref.norm()):.3e} max {float((got - ref).abs().max()):.3e}")
d = (outs["1"].float() - outs["0"].float())
print("split vs no split: max abs diff", float(d.abs().max()), "rel", float(d.norm() / outs["0"].float().norm()),
      "identical outside the tail:", bool(torch.equal(outs["1"][0, :, :11], outs["0"][0, :, :11])))
