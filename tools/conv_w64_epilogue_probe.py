import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
bf = lambda t: t.to(torch.bfloat16)
Cin = Cout = 96; T, H, W, KT = 4, 480, 832, 3
x = bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
wp = bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
bias = torch.randn(Cout, device="cuda")
rf = torch.randn(T, H, W, Cout, device="cuda"); rb = bf(rf)
os.environ["OMH_CONV_TILE"] = "w64"
def t(name, **kw):
    f = lambda: ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 10 * 1e3:.0f} us", flush=True)
for _ in range(2):
    t("bf16 out, no resid"); t("bf16 out, bf16 resid", resid=rb); t("f32 out, no resid", out_f32=True); t("f32 out, f32 resid", resid=rf, out_f32=True)
