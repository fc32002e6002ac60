"""Largest geometry the reference's size table allows: 14B width (d=5120, 40 heads, ffn 13824) at 720x1280, 81 frames
-> latent [16,21,90,160], S = 75 600 tokens (FFN hidden 2.09 GB, just under the kernels' 2 GiB operand limit).
Two layers of the i2v backbone: finite outputs, bit-repeatable, time per block.  GPU box only."""
import importlib, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "omnihuman-1-hack_amd"
model_mod = importlib.import_module(PKG + ".wan.modules.model")
cfgs = importlib.import_module(PKG + ".wan.configs")
dev = torch.device("cuda", 0)
kw = cfgs.dit_kwargs(cfgs.i2v_14B, model_type="i2v", in_dim=36)
kw["num_layers"] = 2
torch.manual_seed(3)
with torch.device(dev):
    m = model_mod.WanModel(**kw)
    torch.nn.init.xavier_uniform_(m.head.head.weight)
m = m.eval().requires_grad_(False)
g = torch.Generator(device=dev).manual_seed(4)
x = torch.randn(16, 21, 90, 160, device=dev, generator=g)
y = torch.randn(20, 21, 90, 160, device=dev, generator=g)
S = 21 * 45 * 80
ctx = [torch.randn(200, 4096, device=dev, generator=g)]
clip = torch.randn(1, 257, 1280, device=dev, generator=g)
t = torch.tensor([500.0], device=dev)
st = m.encode_context(ctx, clip_fea=clip)
out = m([x], t, st, S, y=[y])[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
out2 = m([x], t, st, S, y=[y])[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
d, f, Lc = 5120, 13824, 769
blk = 8 * S * d * d + 4 * S * S * d + 4 * S * d * d + 4 * S * Lc * d + 4 * S * d * f
print(json.dumps({"S": S, "shape": list(out.shape), "finite": bool(torch.isfinite(out).all()),
                  "repeatable": bool(torch.equal(out, out2)), "forward_2_layers_s": round(dt, 3),
                  "tflops": round(2 * blk / dt / 1e12, 1), "absmean": float(out.abs().mean()),
                  "hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
