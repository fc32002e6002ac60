"""Launches only the 480x832 96->96 residual-block convolution (for rocprofv3 --pmc passes); OMH_CONV_TILE picks the kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
Cin = Cout = int(os.environ.get("C", 96)); T, H, W = 4, int(os.environ.get("H", 480)), int(os.environ.get("W", 832))
x = torch.randn(2 + T, H, W, Cin, device="cuda").bfloat16()
wp = (torch.randn(Cout, 27 * Cin, device="cuda") / (27 * Cin) ** 0.5).bfloat16()
bias = torch.randn(Cout, device="cuda")
for _ in range(3):
    ops.conv_cl(x, wp, bias, T, H, W, Cout, 3, 3, 3, pad_h=1, pad_w=1)
torch.cuda.synchronize()
