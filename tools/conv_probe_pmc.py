"""Launches only the 480x832 96->96 residual-block convolution (for rocprofv3 --pmc passes); option CONV_TILE picks the
kernel.  MODE = bf16 (default) | split3 (the fp32-faithful product as a 3 C-channel convolution, rounds 3-4) | pair (the
same product on split-bf16 pairs, round 5: omh_conv_args.pair); TIME=1 prints us per launch instead (no profiler)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
Cin = Cout = int(os.environ.get("C", 96)); T, H, W = 4, int(os.environ.get("H", 480)), int(os.environ.get("W", 832))
mode = os.environ.get("MODE", "bf16")
bias = torch.randn(Cout, device="cuda")
xf = torch.randn(2 + T, H, W, Cin, device="cuda")
wf = torch.randn(Cout * 27, Cin, device="cuda") / (27 * Cin) ** 0.5
if mode == "bf16":
    x, wp, kw = xf.bfloat16(), wf.bfloat16().view(Cout, -1), dict()
elif mode == "split3":
    x, wp, kw = ops.split3(xf, 0), ops.split3(wf, 1).view(Cout, -1), dict(out_f32=True)
else:
    x, wp, kw = ops.split3(xf, 2, Cp=Cin), ops.split3(wf, 2, Cp=Cin).view(Cout, -1), dict(out_f32=True, pair=True)
res = torch.randn(T, H, W, Cout, device="cuda") if mode != "bf16" else None
run = lambda: ops.conv_cl(x, wp, bias, T, H, W, Cout, 3, 3, 3, pad_h=1, pad_w=1, resid=res, **kw)
for _ in range(3):
    run()
torch.cuda.synchronize()
if os.environ.get("TIME"):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    fl = 2.0 * T * H * W * Cout * 27 * Cin * (1 if mode == "bf16" else 3)
    print(f"{mode} C={Cin}: {us:.1f} us per launch, {fl / us / 1e6:.0f} TFLOP/s of executed MFMA work")
