import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
bf = lambda t: t.to(torch.bfloat16)
def t(Cin, Cout, T, H, W, KT):
    x = bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
    wp = bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
    os.environ["OMH_CONV_TILE"] = "w64"
    f = lambda: ops.conv_cl(x, wp, None, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    tiles = -(-T * H * W // 510)
    ns = KT * 3 * (Cin // 32)
    print(f"C{Cin}->{Cout} KT{KT} stages {ns}: {us:.0f} us, {tiles} tiles = {tiles / 256:.2f} rounds -> {us / (tiles / 256):.1f} us per round", flush=True)
for Cin, KT in ((32, 1), (64, 1), (96, 1), (32, 3), (96, 3)):
    t(Cin, 96, 4, 480, 832, KT)
