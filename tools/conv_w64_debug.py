import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
bf = lambda t: t.to(torch.bfloat16)
Cin, Cout, T, H, W, KT = 32, 96, 1, 17, 20, 1
torch.manual_seed(0)
x = bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
w = bf(torch.randn(Cout, Cin, KT, 3, 3, device="cuda") / (Cin * 9) ** 0.5)
wp = w.permute(0, 2, 3, 4, 1).contiguous().view(Cout, -1)
os.environ["OMH_CONV_TILE"] = "w64"
y = ops.conv_cl(x, wp, None, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, out_f32=True)
xin = x.float().permute(3, 0, 1, 2)[None]
def ref(wm):
    return torch.nn.functional.conv3d(torch.nn.functional.pad(xin, (1, 1, 1, 1)), wm.float())[0].permute(1, 2, 3, 0)
full = ref(w)
d = (y - full).abs().sum(-1).view(-1)
bad = (d > 1e-3).nonzero().view(-1).tolist()
print("bad voxels", bad[:20])
for kh in range(3):
    for kw in range(3):
        wz = w.clone(); wz[:, :, :, kh, kw] = 0
        r = ref(wz)
        e = [(y.view(-1, Cout)[v] - r.view(-1, Cout)[v]).abs().max().item() for v in bad[:3]]
        print("zeroed tap kh", kh, "kw", kw, "err at bad voxels", ["%.3f" % t for t in e])
