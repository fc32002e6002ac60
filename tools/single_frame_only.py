"""The batched CFG teacher pair alone (one batch-2 forward at S = 1560 per iteration) for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/single_frame_only.py [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(11)
x = torch.randn(16, 1, 60, 104, device=dev, generator=g)
t = torch.tensor([999.0, 999.0], device=dev)
st = model.encode_context([torch.randn(120, 4096, device=dev, generator=g), torch.randn(40, 4096, device=dev, generator=g)])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    model([x, x], t, st, 1560)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(iters):
    c, u = model([x, x], t, st, 1560)
torch.cuda.synchronize()
print("ms per pair", (time.perf_counter() - t0) / iters * 1e3)
