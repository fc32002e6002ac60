"""Per-shape time of the VAE decode convolutions (one steady-state chunk), for tuning conv_cl_kernel."""
import importlib, os, sys, json, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = "omnihuman-1-hack_amd"
ops = importlib.import_module(pkg + ".ops")
vae_mod = importlib.import_module(pkg + ".wan.modules.vae")
orig = ops.conv_cl
stats = collections.OrderedDict()
def timed(x, w, bias, Tout, Hout, Wout, Cout, KT, KH, KW, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(x, w, bias, Tout, Hout, Wout, Cout, KT, KH, KW, **kw)
    e1.record(); torch.cuda.synchronize()
    key = (tuple(x.shape), Tout, Hout, Wout, Cout, KT, KH, KW, kw.get("stride_hw", 1), bool(kw.get("up2", False)),
           bool(kw.get("out_f32", False)), kw.get("resid") is not None, kw.get("split_n", 0))
    s = stats.setdefault(key, [0, 0.0])
    s[0] += 1; s[1] += e0.elapsed_time(e1)
    return y
vae = vae_mod.WanVAE(vae_pth=None, device="cuda")
z = torch.randn(16, 5, 60, 104, device="cuda")
vae.decode([z[:, :2]])
ops.conv_cl = timed
vae.decode([z]) if os.environ.get("MODE", "decode") == "decode" else vae.encode([torch.rand(3, 17, 480, 832, device="cuda") * 2 - 1])
tot = sum(v[1] for v in stats.values())
for k, (n, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    xs, To, Ho, Wo, Co, KT, KH, KW, st, up, f32, res, sp = k
    fl = 2.0 * To * Ho * Wo * Co * KT * KH * KW * xs[3]
    print(f"x{xs} -> T{To} {Ho}x{Wo} Cout{Co} k{KT}{KH}{KW} s{st} up{int(up)} f32{int(f32)} res{int(res)} split{sp}: "
          f"n={n} {ms:.2f} ms ({100*ms/tot:.1f}%) {fl*n/ms/1e9:.0f} TF/s")
print("total conv ms", tot)
