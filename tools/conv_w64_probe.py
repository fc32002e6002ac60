"""conv_w64.hip's stream kernel against the 8-wave kw-shared kernel: differences per output kind, and timing on the
VAE decoder's layer shapes."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
bf = lambda t: t.to(torch.bfloat16)


def outs(tile, x, wp, bias, rb, rf, T, H, W, Cout, KT):
    ops.set_option("OMH_CONV_TILE", tile)
    kw = dict(pad_h=1, pad_w=1)
    return (ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, resid=rb, **kw),
            ops.conv_cl(x, wp, None, T, H, W, Cout, KT, 3, 3, **kw),
            ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, resid=rf, out_f32=True, **kw),
            ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, out_f32=True, **kw))


def check(Cin, Cout, T, H, W, KT):
    torch.manual_seed(Cin + Cout + H)
    x = bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
    wp = bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    rb, rf = bf(torch.randn(T, H, W, Cout, device="cuda")), torch.randn(T, H, W, Cout, device="cuda")
    got, ref = outs("w64", x, wp, bias, rb, rf, T, H, W, Cout, KT), outs("wide", x, wp, bias, rb, rf, T, H, W, Cout, KT)
    for name, g, r in zip(("bf16+b+r", "bf16", "f32+b+r", "f32+b"), got, ref):
        d = (g.float() - r.float()).abs()
        bad = (d > 0).nonzero()
        print(f"C{Cin}->{Cout} T{T} {H}x{W} KT{KT} {name}: equal {torch.equal(g, r)} max diff {float(d.max()):.3e} n_bad {bad.shape[0]} of {d.numel()}"
              + (f" first bad (t,y,x,c) {bad[0].tolist()} last {bad[-1].tolist()}" if bad.shape[0] else ""), flush=True)


def timeit(Cin, Cout, T, H, W, KT=3, f32=False):
    x = bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
    wp = bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    r = torch.randn(T, H, W, Cout, device="cuda")
    r = r if f32 else bf(r)
    res = {}
    for rnd in range(3):
        for tile in ("wide", "w64"):
            ops.set_option("OMH_CONV_TILE", tile)
            f = lambda: ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, resid=r, out_f32=f32)
            f(); f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                f()
            e.record(); torch.cuda.synchronize()
            res.setdefault(tile, []).append(s.elapsed_time(e) / 5)
    fl = 2.0 * T * H * W * Cout * KT * 9 * Cin
    for tile in ("wide", "w64"):
        ms = sorted(res[tile])[1]
        print(f"time C{Cin}->{Cout} T{T} {H}x{W} {'f32' if f32 else 'bf16'} {tile}: {ms * 1e3:.0f} us {fl / ms / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    if not os.environ.get("TIME_ONLY"):
        for c in ((96, 96, 2, 12, 20, 3), (32, 96, 3, 17, 20, 1), (192, 192, 1, 20, 31, 3), (64, 384, 2, 11, 13, 3), (96, 192, 4, 30, 52, 3)):
            check(*c)
    for f32 in (False, True):
        timeit(96, 96, 4, 480, 832, f32=f32)
        timeit(192, 192, 4, 240, 416, f32=f32)
        timeit(384, 384, 10, 120, 208, f32=f32)
