"""The batched CFG teacher pair at S = 1560 (BASELINE config 1 on the GPU), 20 forwards — for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(11)
x = torch.randn(16, 1, 60, 104, device=dev, generator=g)
t = torch.tensor([999.0, 999.0], device=dev)
ctx = model.encode_context([torch.randn(120, 4096, device=dev, generator=g), torch.randn(40, 4096, device=dev, generator=g)])
for _ in range(22):
    c, u = model([x, x], t, ctx, 1560)
torch.cuda.synchronize()
